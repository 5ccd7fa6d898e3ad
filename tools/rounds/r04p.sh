#!/bin/bash
# round 4, final measurement pass (kernel build 2c4fd9c0ea0238f2) for the tree's kernel build: the full -m gpu suite (with margins), the full PMC counter set
# (SPLINE headline + NEAREST twin), the rocprofv3 kernel stats of the bench command, the bench line itself, cohort trace.
REPO=${GRAFT_REPO_ROOT:-/root/repo}
cd $REPO
O=$REPO/gpurun_out/r04p
mkdir -p $O
export TMPDIR=/tmp
T0=$(date +%s)
el() { echo "[t+$(( $(date +%s) - T0 ))s] $*"; }
echo "== full gpu suite"
NIDREG_MARGINS_OUT=$O/parity_margins.json timeout 1500 python -m pytest tests -q -m gpu --tb=short -p no:cacheprovider > $O/tests_gpu.txt 2>&1; echo "rc=$?"; tail -6 $O/tests_gpu.txt
el "suite done"
echo "== PMC, SPLINE headline"
timeout 900 bash tools/profile_pmc.sh r04p > $O/pmc_spline.log 2>&1; cp gpurun_out/pmc_r04p/summary.txt $O/pmc_summary_fp64.txt
python tools/traffic_from_pmc.py gpurun_out/pmc_r04p $O/traffic.json 10000000 1920 1080 256 fp64 > /dev/null 2>&1; echo "traffic rc=$?"
el "pmc spline done"
echo "== PMC, NEAREST twin"
PMC_DRIVER=run_scene_nearest.py PMC_PASSES="fetch write sq1 sq2" timeout 600 bash tools/profile_pmc.sh r04p_nearest > $O/pmc_nearest.log 2>&1; cp gpurun_out/pmc_r04p_nearest/summary.txt $O/pmc_summary_nearest.txt
python tools/traffic_from_pmc.py gpurun_out/pmc_r04p_nearest $O/pmc_nearest.json 10000000 1920 1080 256 fp64 > /dev/null 2>&1
el "pmc nearest done"
echo "== kernel stats of the bench command"
cd /tmp
timeout 120 rocprofv3 --kernel-trace --stats --output-format csv -d $O/trace -- python $REPO/bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-config-legs > $O/bench_trace.log 2>&1; echo "rc=$?"
cd $REPO
F=$(find $O/trace -name "*kernel_stats.csv" | head -1)
if [ -n "$F" ]; then
  grep -E "Name|nidreg" $F > $O/bench_kernel_stats.csv
  python tools/kernel_stats_json.py $F $O/kernel_stats.json 10000000 1920 1080 256 fp64 "rocprofv3 --kernel-trace --stats -- python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-config-legs"
  cp $O/kernel_stats.json profiles/r04p_kernel_stats.json; cp $O/traffic.json profiles/r04p_traffic.json
fi
rm -rf $O/trace
cd /tmp
timeout 120 rocprofv3 --kernel-trace --stats --output-format csv -d $O/trace_n -- python $REPO/tools/run_scene_nearest.py /tmp/scene.npz 60 > $O/nearest.json 2> /dev/null
cd $REPO
F=$(find $O/trace_n -name "*kernel_stats.csv" | head -1); [ -n "$F" ] && grep -E "Name|nidreg" $F > $O/nearest_kernel_stats.csv; rm -rf $O/trace_n
el "kernel stats done"
echo "== bench (default: every leg)"
timeout 900 python bench.py > $O/bench_line.json 2> $O/bench_err.txt; echo "rc=$?"; cut -c1-400 $O/bench_line.json
el "bench done"
echo "== cohort trace"
python tools/dump_scene_raw.py /tmp/scene.npz /tmp/scene.raw > /dev/null
for K in 2 8; do NIDREG_COHORT_TRACE=1 NIDREG_COHORT=1 OMP_WAIT_POLICY=active timeout 60 tools/omp_pairs.bin 10000000 120 /tmp/scene.raw $K 2>&1 | tee -a $O/cohort_trace.txt; done
el "end"
echo "== concurrent callers, four more times"
for i in 1 2 3 4; do timeout 300 python -m pytest tests/test_concurrent_callers.py -q -m gpu -p no:cacheprovider 2>&1 | tail -n 1; done | tee $O/concurrency_repeats.txt
el "repeats done"
