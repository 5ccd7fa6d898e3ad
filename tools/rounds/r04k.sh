#!/bin/bash
# Round 4, pass k: the gathered entropy tail folded into k_entropy_owned (shards driven by their own threads) -- tests, then the
# one-GPU protocol cost with and without the fold on one box.
set -u
OUT=gpurun_out/r04k
mkdir -p $OUT
T0=$(date +%s)
stamp() { echo "[t+$(( $(date +%s) - T0 ))s] $1"; }
python -m pytest tests/test_sharded_concurrent.py tests/test_gpu_parity.py -q -m gpu -k "own_host_threads or folded or small_tables or shard" > $OUT/tests.txt 2>&1; echo "tests rc=$?"; tail -n 4 $OUT/tests.txt
stamp "tests done"
timeout 300 python tools/shard_cost.py 256 > $OUT/shard_protocol_cost_b256_fold.json 2> $OUT/fold_err.txt; echo rc=$?
NIDREG_SHARD_NO_FOLD=1 timeout 300 python tools/shard_cost.py 256 > $OUT/shard_protocol_cost_b256_nofold.json 2> $OUT/nofold_err.txt; echo rc=$?
python - <<'PY'
import json
for tag in ("fold", "nofold"):
    d = json.load(open(f"gpurun_out/r04k/shard_protocol_cost_b256_{tag}.json"))
    for size, row in d.items():
        print(tag, size, {k: (v["us_per_eval_cost_grad"], v["us_per_eval_cost_only"]) for k, v in row.items()})
PY
stamp "end"
