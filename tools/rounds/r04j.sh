#!/bin/bash
# Round 4, pass j: small tables without an entropy kernel (A/B on one box), then the whole GPU suite on the new chunk tables.
set -u
OUT=gpurun_out/r04j
mkdir -p $OUT
T0=$(date +%s)
stamp() { echo "[t+$(( $(date +%s) - T0 ))s] $1"; }
python -m pytest tests/test_gpu_parity.py -x -q -k "small_tables or value_gradient_histogram" > $OUT/test_new.txt 2>&1; echo "new test rc=$?"; tail -n 3 $OUT/test_new.txt
SIZES=30000,100000,300000,1000000,3000000
python tools/small_cloud_sweep.py 16,32 $SIZES 0 > $OUT/self.jsonl 2> $OUT/self_err.txt; echo rc=$?
NIDREG_NO_SELF_ENTROPY=1 python tools/small_cloud_sweep.py 16,32 $SIZES 0 > $OUT/kentropy.jsonl 2> $OUT/kentropy_err.txt; echo rc=$?
python - <<'PY'
import json
def rows(p):
    return {(r["points"], r["bins"]): r for r in map(json.loads, open(p))}
a, b = rows("gpurun_out/r04j/self.jsonl"), rows("gpurun_out/r04j/kentropy.jsonl")
for k in sorted(a):
    print(k, "no entropy kernel", a[k]["us_per_eval"]["0"], a[k]["kernel_us_rule"], "| k_entropy", b[k]["us_per_eval"]["0"], b[k]["kernel_us_rule"])
PY
stamp "ab done"
python -m pytest tests -x -q -m gpu > $OUT/tests_gpu.txt 2>&1; echo "suite rc=$?"; tail -n 4 $OUT/tests_gpu.txt
cp gpurun_out/parity_margins.json $OUT/parity_margins.json 2>/dev/null
stamp "suite done"
python bench.py > $OUT/bench_line.json 2> $OUT/bench_err.txt; echo "bench rc=$?"; head -c 700 $OUT/bench_line.json; echo
stamp "end"
