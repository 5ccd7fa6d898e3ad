#!/bin/bash
# round 3, GPU pass l: the entropy tail in the gradient kernel's prologue (k_entropy stores partials only) -- same-box A/B
# against the previous build, on cfg 2 and cfg 1; parity tests.
REPO=${GRAFT_REPO_ROOT:-/root/repo}
cd $REPO
O=$REPO/gpurun_out/r03l
mkdir -p $O
export TMPDIR=/tmp
timeout 300 python tools/make_scene_cache.py /tmp/scene.npz > $O/make_scene.log 2>&1
timeout 120 python tools/make_scene_cache.py /tmp/scene_vga.npz pinhole_vga 100000 > /dev/null 2>&1
show() { python3 -c "
import json,sys
d=json.loads(sys.stdin.read()); k=d['kernel_ms']
print('$1 wall=%.4f batch=%.4f whole_ev=%.4f hist=%.4f entropy=%.4f grad=%.4f cost=%r' % (d['wall_ms'], d['wall_batch_ms'], d['whole_eval_event_ms'], k['hist'], k['entropy'], k['grad'], d['last_cost']))"; }
for i in 1 2 3; do for v in prev new; do
  NIDREG_LIB=$REPO/variants/libnidreg_$v.so timeout 200 python tools/run_scene.py /tmp/scene.npz 16 2>&1 | tail -1 | show "cfg2 $v" | tee -a $O/ab.txt
done; done
for i in 1 2; do for v in prev new; do
  NIDREG_LIB=$REPO/variants/libnidreg_$v.so timeout 100 python tools/run_scene.py /tmp/scene_vga.npz 16 fp64 16 2>&1 | tail -1 | show "cfg1 $v" | tee -a $O/ab.txt
done; done
echo "== tests"
timeout 2400 python -m pytest tests -q -m gpu --tb=short -p no:cacheprovider -k "not 50m" > $O/tests_gpu.txt 2>&1; echo "rc=$?"; tail -5 $O/tests_gpu.txt
