#!/bin/bash
# round 3, GPU pass m: the round's final state (entropy tail in the gradient prologue, row sums / Hj behind the histogram, table atan2) -- same passes as r03h.sh:
# --warmup 5 form), rocprofv3 kernel trace + stats of the bench command, PMC passes (traffic / VALU), config table.
REPO=${GRAFT_REPO_ROOT:-/root/repo}
cd $REPO
O=$REPO/gpurun_out/r03m
mkdir -p $O
export TMPDIR=/tmp
echo "== bench (default)"
timeout 900 python bench.py > $O/bench_line.json 2> $O/bench_err.txt; echo "rc=$?"; cut -c1-400 $O/bench_line.json
echo "== bench --steps 20 --warmup 5"
timeout 900 python bench.py --steps 20 --warmup 5 > $O/bench_line_steps20.json 2>> $O/bench_err.txt; echo "rc=$?"; cut -c1-200 $O/bench_line_steps20.json
echo "== kernel trace of the bench command"
cd /tmp
timeout 900 rocprofv3 --kernel-trace --stats --output-format csv -d $O/trace -- python $REPO/bench.py --steps 20 --warmup 5 --no-cpu-baseline > $O/bench_trace.log 2>&1; echo "rc=$?"
cd $REPO
F=$(find $O/trace -name "*kernel_stats.csv" | head -1)
if [ -n "$F" ]; then
  grep -E "Name|nidreg" $F > $O/bench_kernel_stats.csv
  python tools/kernel_stats_json.py $F $O/kernel_stats.json 10000000 1920 1080 256 fp64 "rocprofv3 --kernel-trace --stats -- python bench.py --steps 20 --warmup 5 --no-cpu-baseline"
  python tools/trace_gaps.py $O/trace > $O/kernel_gaps.txt 2>&1; tail -14 $O/kernel_gaps.txt
fi
echo "== PMC passes (torch-free driver)"
timeout 1500 bash tools/profile_pmc.sh r03m fp64 256 0 0 0 > $O/pmc.log 2>&1; tail -3 $O/pmc.log
cp gpurun_out/pmc_r03m/summary.txt $O/pmc_summary_fp64.txt 2>/dev/null
python tools/traffic_from_pmc.py gpurun_out/pmc_r03m $O/traffic.json 10000000 1920 1080 256 fp64 > /dev/null 2>&1; cat $O/traffic.json | head -40
echo "== config table"
timeout 1500 bash tools/config_table.sh > $O/config_table.txt 2>&1; cp gpurun_out/config_table.jsonl $O/config_table.jsonl; cat $O/config_table.txt | cut -c1-300
echo "== full gpu suite"
timeout 2400 python -m pytest tests -q -m gpu --tb=short -p no:cacheprovider > $O/tests_gpu.txt 2>&1; echo "rc=$?"; tail -8 $O/tests_gpu.txt
find $O -name "*.db" -delete; rm -rf $O/trace gpurun_out/pmc_r03m/*/ 2>/dev/null
