#!/bin/bash
# Time libraries under variants/ on one cached 10M-point scene (torch-free driver).
# Usage: run_variants.sh steps "variant:tb variant:tb ..."
REPO=${GRAFT_REPO_ROOT:-/root/repo}
STEPS=${1:-12}
SPECS=${2:-"base:0"}
OUT=$REPO/gpurun_out/variants.jsonl
: > $OUT
python $REPO/tools/make_scene_cache.py /tmp/scene.npz > /tmp/make_scene.log 2>&1
for spec in $SPECS; do
  name=${spec%%:*}; tb=${spec#*:}
  NIDREG_LIB=$REPO/variants/libnidreg_$name.so timeout 120 python $REPO/tools/run_scene.py /tmp/scene.npz $STEPS fp64 256 0 $tb 2>&1 | tail -1 | sed "s/^{/{\"variant\": \"$name\", \"tb\": $tb, /" >> $OUT
done
python3 - <<'PY'
import json,os
p=os.path.join(os.environ.get('GRAFT_REPO_ROOT','/root/repo'),'gpurun_out','variants.jsonl')
for l in open(p):
    try: d=json.loads(l)
    except Exception: print('BAD', l[:200]); continue
    k=d['kernel_ms']
    print(f"{d['variant']:14s} tb={d['tb']:5d} wall={d.get('wall_ms')} total={k.get('total')} hist={k.get('hist')} ent={k.get('entropy')} grad={k.get('grad')} cost={d['last_cost']!r} g0={d['last_grad'][0]!r}")
PY
