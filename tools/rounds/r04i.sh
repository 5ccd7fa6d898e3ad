#!/bin/bash
# Round 4, pass i: the chunk-count rule for small clouds (round_chunks) -- rule against the full round on one box, the NEAREST
# twin's sweep, and the GPU tests that look at chunk tables.
set -u
OUT=gpurun_out/r04i
mkdir -p $OUT
T0=$(date +%s)
stamp() { echo "[t+$(( $(date +%s) - T0 ))s] $1"; }
SIZES=30000,100000,300000,1000000,3000000,10000000
echo "== rule"
python tools/small_cloud_sweep.py 16,64,256 $SIZES 0 > $OUT/rule.jsonl 2> $OUT/rule_err.txt; echo rc=$?
echo "== full round"
NIDREG_FULL_ROUND=1 python tools/small_cloud_sweep.py 16,64,256 $SIZES 0 > $OUT/full_round.jsonl 2> $OUT/full_err.txt; echo rc=$?
stamp "spline done"
echo "== nearest sweep"
python tools/small_cloud_sweep.py 16,256 30000,100000,1000000,3000000 0,64,128,256,512,1024 nearest > $OUT/nearest_sweep.jsonl 2> $OUT/nearest_err.txt; echo rc=$?
stamp "nearest done"
python - <<'PY'
import json
def rows(p):
    return {(r["points"], r["bins"]): r for r in map(json.loads, open(p))}
a, b = rows("gpurun_out/r04i/rule.jsonl"), rows("gpurun_out/r04i/full_round.jsonl")
for k in sorted(a):
    print(k, "rule", a[k]["chunks"]["0"], a[k]["us_per_eval"]["0"], "full", b[k]["chunks"]["0"], b[k]["us_per_eval"]["0"])
for line in open("gpurun_out/r04i/nearest_sweep.jsonl"):
    print(line.strip())
PY
echo "== gpu tests that look at chunk tables"
python -m pytest tests -x -q -m gpu -k "cohort or chunk or multi or concurrent or shard or edge or culling" > $OUT/tests.txt 2>&1; echo rc=$?
tail -n 5 $OUT/tests.txt
stamp "end"
