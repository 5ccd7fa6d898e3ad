#!/bin/bash
# rocprofv3 capture of the default bench command (GPU box).  Kernel trace + stats in one run, PMC
# counters in their own runs (never combined with sys/runtime traces).  Results -> gpurun_out/prof_<tag>/
TAG=${1:-r1}
shift
EXTRA="$@"
REPO=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$REPO/gpurun_out/prof_$TAG
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace -- python $REPO/bench.py --steps 20 --warmup 3 --no-cpu-baseline $EXTRA > $OUT/bench_trace.log 2>&1
echo "trace rc=$?"
rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $OUT/pmc_fetch -- python $REPO/bench.py --steps 4 --warmup 1 --no-cpu-baseline $EXTRA > $OUT/bench_pmc_fetch.log 2>&1
echo "pmc fetch rc=$?"
rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d $OUT/pmc_write -- python $REPO/bench.py --steps 4 --warmup 1 --no-cpu-baseline $EXTRA > $OUT/bench_pmc_write.log 2>&1
echo "pmc write rc=$?"
rocprofv3 --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_INST_LDS SQ_ACTIVE_INST_VALU --kernel-trace --output-format csv -d $OUT/pmc_sq -- python $REPO/bench.py --steps 4 --warmup 1 --no-cpu-baseline $EXTRA > $OUT/bench_pmc_sq.log 2>&1
echo "pmc sq rc=$?"
find $OUT -name "*.csv" | head -30
