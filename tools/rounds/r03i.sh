#!/bin/bash
# round 3, GPU pass i: sweep-rotation experiment (-DNID_EXP_ROTATE): every workgroup starts its chunk at a different iteration
REPO=${GRAFT_REPO_ROOT:-/root/repo}
cd $REPO
O=$REPO/gpurun_out/r03i
mkdir -p $O
timeout 300 python tools/make_scene_cache.py /tmp/scene.npz > $O/make_scene.log 2>&1
for i in 1 2; do for v in base rot1 rot5; do
  NIDREG_LIB=$REPO/variants/libnidreg_$v.so timeout 200 python tools/run_scene.py /tmp/scene.npz 16 2>&1 | tail -1 | python3 -c "
import json,sys
d=json.loads(sys.stdin.read()); k=d['kernel_ms']
print('$v batch=%.4f hist=%.4f entropy=%.4f grad=%.4f total=%.4f cost=%r' % (d['wall_batch_ms'], k['hist'], k['entropy'], k['grad'], k['total'], d['last_cost']))" | tee -a $O/rotate.txt
done; done
