#!/bin/bash
# round 3, GPU pass c: k_fused after the prologue restructure (phi(p) / column term before the second barrier, late
# arguments read where they are used: no SGPR spills in the loops) -- phase timeline, same-box A/B on the BASELINE configs.
REPO=${GRAFT_REPO_ROOT:-/root/repo}
cd $REPO
O=$REPO/gpurun_out/r03c
mkdir -p $O
export TMPDIR=/tmp
timeout 300 python tools/make_scene_cache.py /tmp/scene.npz > $O/make_scene.log 2>&1
echo "== phase timeline of k_fused (stamped build)"
NIDREG_LIB=$REPO/variants/libnidreg_stamp.so timeout 200 python tools/fused_stamps.py /tmp/scene.npz 8 > $O/fused_stamps.txt 2>&1; tail -10 $O/fused_stamps.txt
show() { python3 -c "
import json,sys
for l in sys.stdin:
    try: d=json.loads(l)
    except Exception: print('BAD', l[:300]); continue
    print('$1 fused_env=%s flag=%s wall=%.4f batch=%.4f whole_ev=%.4f k=%s' % (d.get('fused_env'), d['info'].get('fused'), d['wall_ms'], d['wall_batch_ms'], d['whole_eval_event_ms'], d['kernel_ms']))"; }
echo "== A/B cfg2 (10M points, 1080p plumb_bob, 256 bins)"
for i in 1 2 3; do
  for f in 0 1; do NIDREG_FUSED=$f timeout 200 python tools/run_scene.py /tmp/scene.npz 12 2>&1 | tail -1 | tee -a $O/ab_cfg2.jsonl | show cfg2; done
done
echo "== A/B cfg1 (100k points, VGA, 16 bins)"
timeout 120 python tools/make_scene_cache.py /tmp/scene_vga.npz pinhole_vga 100000 > /dev/null 2>&1
for f in 0 1 0 1; do NIDREG_FUSED=$f timeout 100 python tools/run_scene.py /tmp/scene_vga.npz 12 fp64 16 2>&1 | tail -1 | tee -a $O/ab_cfg1.jsonl | show cfg1; done
echo "== A/B cfg5 (50M points, 4K plumb_bob, 256 bins)"
timeout 300 python tools/make_scene_cache.py /tmp/scene_4k.npz pinhole_4k 50000000 > /dev/null 2>&1
for f in 0 1; do NIDREG_FUSED=$f timeout 300 python tools/run_scene.py /tmp/scene_4k.npz 12 2>&1 | tail -1 | tee -a $O/ab_cfg5.jsonl | show cfg5; done
rm -f /tmp/scene_4k.npz
echo "== A/B omnidir 10M (2048x2048, 256 bins)"
timeout 300 python tools/make_scene_cache.py /tmp/scene_om.npz omnidir_2k 10000000 > /dev/null 2>&1
for f in 0 1; do NIDREG_FUSED=$f timeout 300 python tools/run_scene.py /tmp/scene_om.npz 12 2>&1 | tail -1 | tee -a $O/ab_omnidir.jsonl | show omnidir; done
echo "== fused + concurrency tests"
timeout 900 python -m pytest tests/test_fused.py tests/test_concurrent_callers.py -q -m gpu --tb=short -p no:cacheprovider > $O/tests_fused.txt 2>&1; echo "rc=$?"; tail -4 $O/tests_fused.txt
