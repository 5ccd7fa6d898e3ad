#!/bin/bash
# round 3, GPU pass a: the one-launch evaluation (k_fused) -- tests, same-box A/B against the three-kernel route, the full
# -m gpu suite, bench line, kernel trace.  Everything lands in gpurun_out/r03a/.
REPO=${GRAFT_REPO_ROOT:-/root/repo}
cd $REPO
O=$REPO/gpurun_out/r03a
mkdir -p $O
export TMPDIR=/tmp
echo "== fused tests"; timeout 1200 python -m pytest tests/test_fused.py tests/test_concurrent_callers.py -q -m gpu --tb=short -p no:cacheprovider > $O/tests_fused.txt 2>&1; echo "rc=$?"; tail -5 $O/tests_fused.txt
echo "== A/B (torch-free driver, 10M points)"
timeout 300 python tools/make_scene_cache.py /tmp/scene.npz > $O/make_scene.log 2>&1
for i in 1 2; do
  NIDREG_FUSED=0 timeout 200 python tools/run_scene.py /tmp/scene.npz 12 2>&1 | tail -1 >> $O/ab.jsonl
  NIDREG_FUSED=1 timeout 200 python tools/run_scene.py /tmp/scene.npz 12 2>&1 | tail -1 >> $O/ab.jsonl
done
python3 - <<'PY'
import json,os
p=os.path.join(os.environ.get('GRAFT_REPO_ROOT','/root/repo'),'gpurun_out','r03a','ab.jsonl')
for l in open(p):
    try: d=json.loads(l)
    except Exception: print('BAD', l[:300]); continue
    print('fused_env=%s fusedflag=%s wall=%.4f batch=%.4f whole_ev=%.4f k=%s cost=%r' % (d.get('fused_env'), d['info'].get('fused'), d['wall_ms'], d['wall_batch_ms'], d['whole_eval_event_ms'], d['kernel_ms'], d['last_cost']))
PY
echo "== kernel trace (fused, then three kernels)"
cd /tmp
for f in 1 0; do
  NIDREG_FUSED=$f timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/trace_f$f -- python $REPO/tools/run_scene.py /tmp/scene.npz 12 > $O/trace_f$f.log 2>&1
  F=$(find $O/trace_f$f -name "*kernel_stats.csv" | head -1)
  [ -n "$F" ] && cp $F $O/kernel_stats_f$f.csv && python3 - "$F" <<'PY'
import csv,sys
for r in csv.DictReader(open(sys.argv[1])):
    if 'nidreg' in r['Name']: print(f"{r['Name'][:90]:90s} calls={r['Calls']:>5s} avg_us={float(r['AverageNs'])/1e3:9.2f} min={float(r['MinNs'])/1e3:8.2f} max={float(r['MaxNs'])/1e3:8.2f}")
PY
done
cd $REPO
echo "== bench"
timeout 600 python bench.py > $O/bench_line.json 2> $O/bench_err.txt; echo "rc=$?"; cut -c1-600 $O/bench_line.json
echo "== full gpu suite"
timeout 2400 python -m pytest tests -q -m gpu --tb=short -p no:cacheprovider > $O/tests_gpu.txt 2>&1; echo "rc=$?"; tail -15 $O/tests_gpu.txt
find $O -name "*.db" -delete; find $O -type d -name "trace_f*" -exec du -sh {} \;
