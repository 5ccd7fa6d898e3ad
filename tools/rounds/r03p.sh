#!/bin/bash
# round 3, GPU pass p (the round's last GPU seconds): kernel trace of the bench command for the kernel build of r03o.sh,
# on a box without the OMP_WAIT_POLICY=active that r03o.sh leaked into its bench / trace steps.
REPO=${GRAFT_REPO_ROOT:-/root/repo}
O=$REPO/gpurun_out/r03p
mkdir -p $O
export TMPDIR=/tmp
cd /tmp
timeout 60 rocprofv3 --kernel-trace --stats --output-format csv -d $O/trace -- python $REPO/bench.py --steps 20 --warmup 5 --no-cpu-baseline > $O/bench_trace.log 2>&1; echo "rc=$?"
cd $REPO
F=$(find $O/trace -name "*kernel_stats.csv" | head -1)
if [ -n "$F" ]; then
  grep -E "Name|nidreg" $F > $O/bench_kernel_stats.csv
  python tools/kernel_stats_json.py $F $O/kernel_stats.json 10000000 1920 1080 256 fp64 "rocprofv3 --kernel-trace --stats -- python bench.py --steps 20 --warmup 5 --no-cpu-baseline"
  cp $O/kernel_stats.json profiles/r03p_kernel_stats.json
fi
rm -rf $O/trace
timeout 40 python bench.py --steps 20 --warmup 5 --no-cpu-baseline > $O/bench_line_steps20_nocpu.json 2> $O/bench_err.txt; echo "rc=$?"; cut -c1-200 $O/bench_line_steps20_nocpu.json
