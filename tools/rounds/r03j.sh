#!/bin/bash
# round 3, GPU pass j: table-based atan2 in the SPLINE kernels (fisheye / equirectangular): parity tests + the config table
REPO=${GRAFT_REPO_ROOT:-/root/repo}
cd $REPO
O=$REPO/gpurun_out/r03j
mkdir -p $O
export TMPDIR=/tmp
echo "== parity tests touching the projections"
timeout 1500 python -m pytest tests/test_gpu_parity.py tests/test_reference_golden.py tests/test_fused.py tests/test_cxx_dropin.py -q -m gpu --tb=short -p no:cacheprovider -k "not 50m and not 10m and not headline" > $O/tests.txt 2>&1; echo "rc=$?"; tail -6 $O/tests.txt
echo "== config table"
timeout 1500 bash tools/config_table.sh > $O/config_table.txt 2>&1; cp gpurun_out/config_table.jsonl $O/config_table.jsonl; cut -c1-200 $O/config_table.txt
python3 - <<'PY'
import json,os
p=os.path.join(os.environ.get('GRAFT_REPO_ROOT','/root/repo'),'gpurun_out','r03j','config_table.jsonl')
for l in open(p):
    d=json.loads(l); c=d['config']; k=d['roofline']['kernel_ms_events']['three_kernel_route']
    print(c['points'], c['camera_model'], d['value'], d['ms_per_step'], 'hist %.4f grad %.4f' % (k['hist'], k['grad']), c['layout'].get('num_chunks'))
PY
