#!/bin/bash
# round 3, GPU pass q (the round's last GPU seconds): chunk tables that fit one round (split_groups) -- the single grid at
# 2 / 4 / 8 pairs again, per-pair launches at 2 pairs beside it, then smoke().
REPO=${GRAFT_REPO_ROOT:-/root/repo}
cd $REPO
O=$REPO/gpurun_out/r03q
mkdir -p $O
export TMPDIR=/tmp
timeout 40 python tools/make_scene_cache.py /tmp/scene.npz > $O/make_scene.log 2>&1
python tools/dump_scene_raw.py /tmp/scene.npz /tmp/scene.raw > /dev/null
for K in 2 4 8; do
  NIDREG_MULTI_GRID_MIN=2 OMP_WAIT_POLICY=active timeout 20 tools/omp_pairs.bin 10000000 120 /tmp/scene.raw $K | tee -a $O/omp_pairs.jsonl
done
NIDREG_NO_MULTI_GRID=1 OMP_WAIT_POLICY=active timeout 20 tools/omp_pairs.bin 10000000 120 /tmp/scene.raw 2 | tee -a $O/omp_pairs.jsonl
timeout 30 python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -4 | tee $O/smoke.txt
