#!/bin/bash
# round 4, first GPU pass (written at the end of round 3, whose GPU budget ended before these could run):
# 1. the full -m gpu suite on the tree (last full run: r03m, before the multi-pair kernels / chunk rule changed)
# 2. the chunk rule on a view-culled cloud, old rule against split_groups (tools/culled_cloud_ab.py)
# 3. bench (default), config table
# Keep OMP_WAIT_POLICY out of the environment of anything but omp_pairs (r03o.sh's mistake).
REPO=${GRAFT_REPO_ROOT:-/root/repo}
cd $REPO
O=$REPO/gpurun_out/r04a
mkdir -p $O
export TMPDIR=/tmp
T0=$(date +%s)
el() { echo "[t+$(( $(date +%s) - T0 ))s] $*"; }
echo "== full gpu suite"
timeout 1200 python -m pytest tests -q -m gpu --tb=short -p no:cacheprovider > $O/tests_gpu.txt 2>&1; echo "rc=$?"; tail -6 $O/tests_gpu.txt
el "suite done"
echo "== view-culled cloud, chunk rule A/B"
timeout 300 python tools/culled_cloud_ab.py 10000000 pinhole_1080p | tee $O/culled_cloud_ab.json
timeout 300 python tools/culled_cloud_ab.py 10000000 equirect_2k | tee -a $O/culled_cloud_ab.json
el "culled A/B done"
echo "== bench (default)"
timeout 600 python bench.py > $O/bench_line.json 2> $O/bench_err.txt; echo "rc=$?"; cut -c1-300 $O/bench_line.json
el "bench done"
echo "== multi-pair routes"
timeout 120 python tools/make_scene_cache.py /tmp/scene.npz > $O/make_scene.log 2>&1
python tools/dump_scene_raw.py /tmp/scene.npz /tmp/scene.raw > /dev/null
for K in 2 3 4 8; do
  OMP_WAIT_POLICY=active timeout 60 tools/omp_pairs.bin 10000000 120 /tmp/scene.raw $K | tee -a $O/omp_pairs.jsonl
  NIDREG_NO_MULTI_GRID=1 OMP_WAIT_POLICY=active timeout 60 tools/omp_pairs.bin 10000000 120 /tmp/scene.raw $K | tee -a $O/omp_pairs.jsonl
done
el "omp_pairs done"
echo "== config table"
timeout 900 bash tools/config_table.sh > $O/config_table.txt 2>&1; cp gpurun_out/config_table.jsonl $O/config_table.jsonl; cut -c1-300 $O/config_table.txt
el "end"
