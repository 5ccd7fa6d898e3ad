#!/bin/bash
# rocprofv3 --kernel-trace --stats of the torch-free driver.  Usage: profile_trace.sh <tag> [run_scene args...]
TAG=${1:-r1}
shift
ARGS="$@"
REPO=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$REPO/gpurun_out/trace_$TAG
mkdir -p $OUT
[ -f /tmp/scene.npz ] || python $REPO/tools/make_scene_cache.py /tmp/scene.npz > $OUT/make_scene.log 2>&1
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/t -- python $REPO/tools/run_scene.py /tmp/scene.npz 30 $ARGS > $OUT/run.log 2>&1
echo "rc=$?"; tail -1 $OUT/run.log | cut -c1-400
F=$(find $OUT -name "*kernel_stats.csv" | head -1)
python3 - "$F" <<'PY'
import csv,sys
for r in csv.DictReader(open(sys.argv[1])):
    print(f"{r['Name'][:70]:70s} calls={r['Calls']:>5s} avg_us={float(r['AverageNs'])/1e3:9.2f} min={float(r['MinNs'])/1e3:8.2f} max={float(r['MaxNs'])/1e3:8.2f}")
PY
