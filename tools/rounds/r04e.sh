#!/bin/bash
# round 4, fifth GPU pass: cohorts (the unchanged OpenMP caller), targeted tests of everything changed since r04d.
REPO=${GRAFT_REPO_ROOT:-/root/repo}
cd $REPO
O=$REPO/gpurun_out/r04e
mkdir -p $O
export TMPDIR=/tmp
T0=$(date +%s)
el() { echo "[t+$(( $(date +%s) - T0 ))s] $*"; }
echo "== targeted tests"
NIDREG_MARGINS_OUT=$O/parity_margins_targeted.json timeout 900 python -m pytest tests/test_gpu_parity.py tests/test_concurrent_callers.py -q -m gpu --tb=short -p no:cacheprovider \
  -k "cohort or chunks_across or submit_wait or nearest_fast or deterministic_and_tiling or multi_pair_single_grid or concurrent or value_gradient or headline" > $O/tests_targeted.txt 2>&1; echo "rc=$?"; tail -8 $O/tests_targeted.txt
el "targeted done"
echo "== omp_pairs: own tables / cohort"
timeout 120 python tools/make_scene_cache.py /tmp/scene.npz > $O/make_scene.log 2>&1
python tools/dump_scene_raw.py /tmp/scene.npz /tmp/scene.raw > /dev/null
for K in 2 4 8; do
  OMP_WAIT_POLICY=active timeout 60 tools/omp_pairs.bin 10000000 120 /tmp/scene.raw $K | tee -a $O/omp_pairs_own.jsonl
  NIDREG_COHORT=1 OMP_WAIT_POLICY=active timeout 60 tools/omp_pairs.bin 10000000 120 /tmp/scene.raw $K | tee -a $O/omp_pairs_cohort.jsonl
done
el "omp_pairs done"
echo "== headline with the final entropy shape"
timeout 120 python tools/run_scene.py /tmp/scene.npz 30 | tee $O/run_scene.json | cut -c1-500
el "end"
