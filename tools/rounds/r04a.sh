#!/bin/bash
# round 4, first GPU pass: the tree as round 3 left it (kernel build c356e3d1277d4670) plus the parity-margin recorder.
# 1. the full -m gpu suite; every oracle comparison records its margin (tests/parity.py -> parity_margins.json)
# 2. the full PMC counter set of THIS build on the headline workload (SPLINE) and on the NEAREST twin
# 3. bench (default), the chunk rule on a view-culled cloud (tools/culled_cloud_ab.py)
REPO=${GRAFT_REPO_ROOT:-/root/repo}
cd $REPO
O=$REPO/gpurun_out/r04a
mkdir -p $O
export TMPDIR=/tmp
T0=$(date +%s)
el() { echo "[t+$(( $(date +%s) - T0 ))s] $*"; }
echo "== full gpu suite"
NIDREG_MARGINS_OUT=$O/parity_margins.json timeout 1200 python -m pytest tests -q -m gpu --tb=short -p no:cacheprovider > $O/tests_gpu.txt 2>&1; echo "rc=$?"; tail -6 $O/tests_gpu.txt
el "suite done"
echo "== PMC, SPLINE headline"
timeout 900 bash tools/profile_pmc.sh r04a > $O/pmc_spline.log 2>&1; cp gpurun_out/pmc_r04a/summary.txt $O/pmc_summary_fp64.txt; head -70 $O/pmc_summary_fp64.txt
el "pmc spline done"
echo "== PMC, NEAREST twin"
PMC_DRIVER=run_scene_nearest.py PMC_PASSES="fetch write sq1 sq2" timeout 600 bash tools/profile_pmc.sh r04a_nearest > $O/pmc_nearest.log 2>&1; cp gpurun_out/pmc_r04a_nearest/summary.txt $O/pmc_summary_nearest.txt; head -40 $O/pmc_summary_nearest.txt
el "pmc nearest done"
echo "== bench (default)"
timeout 600 python bench.py > $O/bench_line.json 2> $O/bench_err.txt; echo "rc=$?"; cut -c1-400 $O/bench_line.json
el "bench done"
echo "== view-culled cloud, chunk rule A/B"
timeout 300 python tools/culled_cloud_ab.py 10000000 pinhole_1080p | tee $O/culled_cloud_ab.json
el "end"
