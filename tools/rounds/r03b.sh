#!/bin/bash
# round 3, GPU pass b: two-level grid barrier in k_fused (phase timeline + same-box A/B), the column-owned sharding (tests,
# protocol cost).  Everything lands in gpurun_out/r03b/.
REPO=${GRAFT_REPO_ROOT:-/root/repo}
cd $REPO
O=$REPO/gpurun_out/${RTAG:-r03b}
mkdir -p $O
export TMPDIR=/tmp
timeout 300 python tools/make_scene_cache.py /tmp/scene.npz > $O/make_scene.log 2>&1
echo "== phase timeline of k_fused (stamped build)"
NIDREG_LIB=$REPO/variants/libnidreg_stamp.so timeout 200 python tools/fused_stamps.py /tmp/scene.npz 8 > $O/fused_stamps.txt 2>&1; cat $O/fused_stamps.txt | tail -12
echo "== A/B (torch-free driver, 10M points)"
for i in 1 2; do
  NIDREG_FUSED=0 timeout 200 python tools/run_scene.py /tmp/scene.npz 12 2>&1 | tail -1 >> $O/ab.jsonl
  NIDREG_FUSED=1 timeout 200 python tools/run_scene.py /tmp/scene.npz 12 2>&1 | tail -1 >> $O/ab.jsonl
done
python3 - <<'PY'
import json,os
p=os.path.join(os.environ.get('GRAFT_REPO_ROOT','/root/repo'),'gpurun_out','r03b','ab.jsonl')
for l in open(p):
    try: d=json.loads(l)
    except Exception: print('BAD', l[:300]); continue
    print('fused_env=%s fusedflag=%s wall=%.4f batch=%.4f whole_ev=%.4f k=%s' % (d.get('fused_env'), d['info'].get('fused'), d['wall_ms'], d['wall_batch_ms'], d['whole_eval_event_ms'], d['kernel_ms']))
PY
echo "== cfg1 (100k points, VGA, 16 bins) A/B"
timeout 120 python tools/make_scene_cache.py /tmp/scene_vga.npz pinhole_vga 100000 > /dev/null 2>&1
for f in 0 1; do NIDREG_FUSED=$f timeout 100 python tools/run_scene.py /tmp/scene_vga.npz 12 fp64 16 2>&1 | tail -1 | python3 -c "
import json,sys
d=json.loads(sys.stdin.read()); print('cfg1 fused_env=%s flag=%s wall=%.4f batch=%.4f whole_ev=%.4f k=%s' % (d.get('fused_env'), d['info'].get('fused'), d['wall_ms'], d['wall_batch_ms'], d['whole_eval_event_ms'], d['kernel_ms']))"; done
echo "== sharding + fused tests"
timeout 1500 python -m pytest tests/test_gpu_parity.py tests/test_sharded_concurrent.py tests/test_fused.py tests/test_concurrent_callers.py tests/test_parallel_gloo.py tests/test_bench_launch.py -q -m gpu --tb=short -p no:cacheprovider -k "shard or devices or fused or concurrent or bench or gloo" > $O/tests_shard.txt 2>&1; echo "rc=$?"; tail -25 $O/tests_shard.txt
echo "== protocol cost of the sharded evaluation (one GPU, co-located shards on worker threads)"
timeout 400 python tools/shard_cost.py 256 > $O/shard_protocol_cost_b256.json 2> $O/shard_cost_256.err; cat $O/shard_protocol_cost_b256.json
timeout 400 python tools/shard_cost.py 16 > $O/shard_protocol_cost_b16.json 2> $O/shard_cost_16.err; cat $O/shard_protocol_cost_b16.json
