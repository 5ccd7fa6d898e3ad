#!/bin/bash
# One bench line per BASELINE.json config that fits one GPU (configs 3-5 run un-sharded / one pair).
REPO=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$REPO/gpurun_out/config_table.jsonl
: > $OUT
run() { python $REPO/bench.py --steps 20 --warmup 3 --cpu-sample 500000 "$@" 2>/dev/null | tail -1 >> $OUT; }
run --points 100000 --camera pinhole_vga --bins 16
run --points 10000000 --camera pinhole_1080p --bins 256
run --points 10000000 --camera equirect_2k --bins 256
run --points 10000000 --camera omnidir_2k --bins 256
run --points 5000000 --camera fisheye_1080p --bins 256
run --points 50000000 --camera pinhole_4k --bins 256
python3 - <<'PY'
import json,os
p=os.path.join(os.environ.get('GRAFT_REPO_ROOT','/root/repo'),'gpurun_out','config_table.jsonl')
for line in open(p):
    line=line.strip()
    if not line: continue
    d=json.loads(line)
    c=d['config']; r=d['roofline'] or {}; cb=d.get('cpu_baseline') or {}
    print(c['points'], c['camera_model'], c['image'], c['bins'], 'evals/s', d['value'], 'ms', d['ms_per_step'], 'frac', r.get('frac'), 'eval_frac', r.get('eval_frac'), 'cpu', cb.get('value'), 'gen', cb.get('generous_value'), 'setup', c['setup_s'], d.get('other_entry_points'))
PY
