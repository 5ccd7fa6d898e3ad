#!/bin/bash
# round 3, GPU pass e: dispatch-slot weights of the chunk tables (part-major order; NIDREG_SLOT_WEIGHTS_*): kernel times of the
# three-kernel route on the headline workload for a sweep of weights.
REPO=${GRAFT_REPO_ROOT:-/root/repo}
cd $REPO
O=$REPO/gpurun_out/r03e
mkdir -p $O
export TMPDIR=/tmp
timeout 300 python tools/make_scene_cache.py /tmp/scene.npz > $O/make_scene.log 2>&1
run() { # label, hist weights, grad weights
  NIDREG_SLOT_WEIGHTS_HIST=$2 NIDREG_SLOT_WEIGHTS_GRAD=$3 timeout 200 python tools/run_scene.py /tmp/scene.npz 16 2>&1 | tail -1 | python3 -c "
import json,sys
d=json.loads(sys.stdin.read()); k=d['kernel_ms']
print('$1 hist_w=$2 grad_w=$3 batch=%.4f hist=%.4f entropy=%.4f grad=%.4f total=%.4f' % (d['wall_batch_ms'], k['hist'], k['entropy'], k['grad'], k['total']))" | tee -a $O/slot_weights.txt
}
run base 1,1 1,1,1,1
run h03 1.03,0.97 1,1,1,1
run h05 1.05,0.95 1,1,1,1
run h07 1.07,0.93 1,1,1,1
run h10 1.10,0.90 1,1,1,1
run g03 1,1 1.03,1.01,0.99,0.97
run g05 1,1 1.05,1.017,0.983,0.95
run g07 1,1 1.07,1.023,0.977,0.93
run g10 1,1 1.10,1.033,0.967,0.90
run gA 1,1 1.06,1.00,1.00,0.94
run gB 1,1 1.08,0.99,0.97,0.96
run base2 1,1 1,1,1,1
