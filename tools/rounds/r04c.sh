#!/bin/bash
# round 4, third GPU pass: looped kernels after the register / prefetch fixes, submit / wait, the NEAREST fast tier.
REPO=${GRAFT_REPO_ROOT:-/root/repo}
cd $REPO
O=$REPO/gpurun_out/r04c
mkdir -p $O
export TMPDIR=/tmp
T0=$(date +%s)
el() { echo "[t+$(( $(date +%s) - T0 ))s] $*"; }
echo "== targeted tests"
NIDREG_MARGINS_OUT=$O/parity_margins_targeted.json timeout 900 python -m pytest tests/test_gpu_parity.py tests/test_calibration.py tests/test_concurrent_callers.py -q -m gpu --tb=short -p no:cacheprovider \
  -k "chunks_across or submit_wait or nearest or deterministic_and_tiling or multi_pair_single_grid or in_library_sharding or outliers_and_padding or odd_bin or calibration or concurrent" > $O/tests_targeted.txt 2>&1; echo "rc=$?"; tail -8 $O/tests_targeted.txt
el "targeted done"
echo "== culled cloud A/B"
timeout 300 python tools/culled_cloud_ab.py 10000000 pinhole_1080p 6.0 | tee $O/culled_cloud_ab.jsonl
timeout 300 python tools/culled_cloud_ab.py 10000000 pinhole_1080p 3.0 | tee -a $O/culled_cloud_ab.jsonl
timeout 300 python tools/culled_cloud_ab.py 10000000 equirect_2k 6.0 | tee -a $O/culled_cloud_ab.jsonl
el "culled done"
echo "== NEAREST twin: fast tier vs exact tier only"
timeout 120 python tools/make_scene_cache.py /tmp/scene.npz > $O/make_scene.log 2>&1
cd /tmp
timeout 200 rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof_nearest_fast -- python $REPO/tools/run_scene_nearest.py /tmp/scene.npz 60 > $O/nearest_fast.json 2> $O/nearest_fast.err; cat $O/nearest_fast.json | cut -c1-300
NIDREG_NEAREST_EXACT=1 timeout 200 rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof_nearest_exact -- python $REPO/tools/run_scene_nearest.py /tmp/scene.npz 60 > $O/nearest_exact.json 2> $O/nearest_exact.err; cat $O/nearest_exact.json | cut -c1-300
cd $REPO
for d in prof_nearest_fast prof_nearest_exact; do f=$(find $O/$d -name "*kernel_stats.csv" | head -1); echo "-- $d"; [ -n "$f" ] && head -6 "$f" | cut -c1-200; done
el "nearest done"
echo "== bench (default, all legs)"
timeout 900 python bench.py > $O/bench_line.json 2> $O/bench_err.txt; echo "rc=$?"; python - <<'PY'
import json
d=json.load(open("gpurun_out/r04c/bench_line.json"))
print({k: d[k] for k in ("value","ms_per_step")}, d.get("pipelined"), d.get("culled"))
print(d.get("configs"))
print(d["roofline"]["kernel_ms_events"], d["other_entry_points"])
PY
tail -3 $O/bench_err.txt
el "end"
