#!/bin/bash
# round 3, GPU pass o: the multi-pair kernels with GLOBAL instead of FLAT memory instructions (nid_multi.hpp, as_global):
# single grid against per-pair launches at 2 / 4 / 8 pairs, the tests of the multi-pair / fused / concurrent routes, and
# the stamped passes for the new kernel build (bench, kernel trace, PMC traffic), most important first.
REPO=${GRAFT_REPO_ROOT:-/root/repo}
cd $REPO
O=$REPO/gpurun_out/r03o
mkdir -p $O
export TMPDIR=/tmp
T0=$(date +%s)
el() { echo "[t+$(( $(date +%s) - T0 ))s] $*"; }
timeout 300 python tools/make_scene_cache.py /tmp/scene.npz > $O/make_scene.log 2>&1
python tools/dump_scene_raw.py /tmp/scene.npz /tmp/scene.raw > /dev/null
el "scene ready"
# (as run, this script exported OMP_WAIT_POLICY=active here for everything below: its bench / trace steps ran beside spinning OpenMP
# threads and their numbers were discarded -- r03p.sh repeated the trace; the variable now only reaches omp_pairs)
for K in 2 4 8; do
  echo "single grid, $K pairs";  NIDREG_MULTI_GRID_MIN=2 OMP_WAIT_POLICY=active timeout 200 tools/omp_pairs.bin 10000000 120 /tmp/scene.raw $K | tee -a $O/omp_pairs.jsonl
  echo "per-pair launches, $K pairs"; NIDREG_NO_MULTI_GRID=1 OMP_WAIT_POLICY=active timeout 200 tools/omp_pairs.bin 10000000 120 /tmp/scene.raw $K | tee -a $O/omp_pairs.jsonl
done
el "omp_pairs done"
echo "== tests of the multi-pair routes (default threshold, then single grid from two pairs)"
timeout 400 python -m pytest tests/test_gpu_parity.py tests/test_concurrent_callers.py -q -m gpu --tb=short -p no:cacheprovider -k "multi or concurrent or smoke" > $O/tests_multi.txt 2>&1; echo "rc=$?"; tail -3 $O/tests_multi.txt
NIDREG_MULTI_GRID_MIN=2 timeout 400 python -m pytest tests/test_gpu_parity.py -q -m gpu --tb=short -p no:cacheprovider -k "multi" > $O/tests_multi_min2.txt 2>&1; echo "rc=$?"; tail -3 $O/tests_multi_min2.txt
el "multi tests done"
echo "== kernel trace of the bench command"
cd /tmp
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $O/trace -- python $REPO/bench.py --steps 20 --warmup 5 --no-cpu-baseline > $O/bench_trace.log 2>&1; echo "rc=$?"
cd $REPO
F=$(find $O/trace -name "*kernel_stats.csv" | head -1)
if [ -n "$F" ]; then
  grep -E "Name|nidreg" $F > $O/bench_kernel_stats.csv
  python tools/kernel_stats_json.py $F $O/kernel_stats.json 10000000 1920 1080 256 fp64 "rocprofv3 --kernel-trace --stats -- python bench.py --steps 20 --warmup 5 --no-cpu-baseline"
fi
el "trace done"
echo "== PMC passes (torch-free driver): fetch, write, instruction counts"
PMC_PASSES="fetch write sq1" timeout 600 bash tools/profile_pmc.sh r03o fp64 256 0 0 0 > $O/pmc.log 2>&1; tail -3 $O/pmc.log
cp gpurun_out/pmc_r03o/summary.txt $O/pmc_summary_fp64.txt 2>/dev/null
python tools/traffic_from_pmc.py gpurun_out/pmc_r03o $O/traffic.json 10000000 1920 1080 256 fp64 > /dev/null 2>&1; head -c 600 $O/traffic.json
el "pmc done"
echo "== bench (default, then --steps 20 --warmup 5) with this build's stamped profiles in place"
cp $O/traffic.json profiles/r03o_traffic.json; cp $O/kernel_stats.json profiles/r03o_kernel_stats.json
timeout 600 python bench.py > $O/bench_line.json 2> $O/bench_err.txt; echo "rc=$?"; cut -c1-400 $O/bench_line.json
el "bench done"
timeout 600 python bench.py --steps 20 --warmup 5 > $O/bench_line_steps20.json 2>> $O/bench_err.txt; echo "rc=$?"; cut -c1-300 $O/bench_line_steps20.json
el "bench 2 done"
echo "== more of the gpu suite, as far as the time goes (fused route, calibration, sharding first)"
timeout ${SUITE_SECONDS:-240} python -m pytest tests/test_fused.py tests/test_calibration.py tests/test_sharded_concurrent.py tests/test_gpu_parity.py -q -m gpu --tb=short -p no:cacheprovider -x > $O/tests_more.txt 2>&1; echo "rc=$?"; tail -4 $O/tests_more.txt
el "end"
find $O -name "*.db" -delete; rm -rf $O/trace gpurun_out/pmc_r03o/*/ 2>/dev/null
