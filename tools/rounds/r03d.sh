#!/bin/bash
# round 3, GPU pass d: where the workgroup spread comes from (stamped k_fused); the reference's multi-bag calling pattern on
# ONE GPU -- OpenMP callers vs nidreg_eval_multi (single grid / per-pair launches) on the same scene clouds; new parity tests.
REPO=${GRAFT_REPO_ROOT:-/root/repo}
cd $REPO
O=$REPO/gpurun_out/r03d
mkdir -p $O
export TMPDIR=/tmp
timeout 300 python tools/make_scene_cache.py /tmp/scene.npz > $O/make_scene.log 2>&1
echo "== phase timeline + spread"
NIDREG_FUSED=1 NIDREG_LIB=$REPO/variants/libnidreg_stamp.so timeout 200 python tools/fused_stamps.py /tmp/scene.npz 10 > $O/fused_stamps.txt 2>&1; tail -14 $O/fused_stamps.txt
echo "== multi-pair patterns on one GPU (scene clouds, 10M points in all)"
python tools/dump_scene_raw.py /tmp/scene.npz /tmp/scene.raw > /dev/null
export OMP_WAIT_POLICY=active
for mode in grid perpair; do
  if [ $mode = perpair ]; then export NIDREG_NO_MULTI_GRID=1; else unset NIDREG_NO_MULTI_GRID; fi
  timeout 600 tools/omp_pairs.bin 10000000 150 /tmp/scene.raw >> $O/omp_pairs_scene.jsonl 2>&1
done
unset NIDREG_NO_MULTI_GRID
timeout 600 tools/omp_pairs.bin 10000000 150 >> $O/omp_pairs_random.jsonl 2>&1
NIDREG_NO_MULTI_GRID=1 timeout 600 tools/omp_pairs.bin 10000000 150 >> $O/omp_pairs_random.jsonl 2>&1
cat $O/omp_pairs_scene.jsonl $O/omp_pairs_random.jsonl
echo "== new parity tests"
timeout 900 python -m pytest tests/test_gpu_parity.py -q -m gpu --tb=short -p no:cacheprovider -k "boundar or multi_pair or multi_nid or thread" > $O/tests_new.txt 2>&1; echo "rc=$?"; tail -5 $O/tests_new.txt
