#!/bin/bash
# round 3, GPU pass f: kernel trace of the multi-pair single-grid evaluation (2 pairs x 5M points) against per-pair launches
REPO=${GRAFT_REPO_ROOT:-/root/repo}
cd $REPO
O=$REPO/gpurun_out/r03f
mkdir -p $O
export TMPDIR=/tmp
timeout 300 python tools/make_scene_cache.py /tmp/scene.npz > $O/make_scene.log 2>&1
python tools/dump_scene_raw.py /tmp/scene.npz /tmp/scene.raw > /dev/null
cd /tmp
for mode in grid perpair; do
  if [ $mode = perpair ]; then export NIDREG_NO_MULTI_GRID=1; else unset NIDREG_NO_MULTI_GRID; fi
  for k in 2 8; do
    timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/trace_${mode}_$k -- $REPO/tools/omp_pairs.bin 10000000 60 /tmp/scene.raw $k > $O/trace_${mode}_$k.log 2>&1
    tail -1 $O/trace_${mode}_$k.log | cut -c1-300
    F=$(find $O/trace_${mode}_$k -name "*kernel_stats.csv" | head -1)
    [ -n "$F" ] && cp $F $O/kernel_stats_${mode}_$k.csv && python3 - "$F" <<'PY'
import csv,sys
for r in csv.DictReader(open(sys.argv[1])):
    if 'nidreg::k_' in r['Name'] and 'build' not in r['Name']: print(f"   {r['Name'][:100]:100s} calls={r['Calls']:>5s} avg_us={float(r['AverageNs'])/1e3:9.2f}")
PY
    T=$(find $O/trace_${mode}_$k -name "*kernel_trace.csv" | head -1)
    [ -n "$T" ] && python3 $REPO/tools/trace_gaps.py $O/trace_${mode}_$k 2>/dev/null | tail -12
  done
done
find $O -name "*.db" -delete
