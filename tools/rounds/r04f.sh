#!/bin/bash
# round 4, sixth GPU pass: cohort rendezvous (share tables + one grid for the callers that arrive together)
REPO=${GRAFT_REPO_ROOT:-/root/repo}
cd $REPO
O=$REPO/gpurun_out/r04f
mkdir -p $O
export TMPDIR=/tmp
T0=$(date +%s)
el() { echo "[t+$(( $(date +%s) - T0 ))s] $*"; }
echo "== targeted tests"
timeout 900 python -m pytest tests/test_gpu_parity.py tests/test_concurrent_callers.py -q -m gpu --tb=short -p no:cacheprovider -k "cohort or multi_pair_single_grid or concurrent or sharding" > $O/tests_targeted.txt 2>&1; echo "rc=$?"; tail -8 $O/tests_targeted.txt
el "targeted done"
echo "== omp_pairs: own tables / cohort with rendezvous"
timeout 120 python tools/make_scene_cache.py /tmp/scene.npz > $O/make_scene.log 2>&1
python tools/dump_scene_raw.py /tmp/scene.npz /tmp/scene.raw > /dev/null
for K in 2 4 8; do
  OMP_WAIT_POLICY=active timeout 60 tools/omp_pairs.bin 10000000 120 /tmp/scene.raw $K | tee -a $O/omp_pairs_own.jsonl
  NIDREG_COHORT=1 OMP_WAIT_POLICY=active timeout 60 tools/omp_pairs.bin 10000000 120 /tmp/scene.raw $K | tee -a $O/omp_pairs_cohort.jsonl
  NIDREG_COHORT=1 timeout 60 tools/omp_pairs.bin 10000000 120 /tmp/scene.raw $K | tee -a $O/omp_pairs_cohort_passive.jsonl
done
el "end"
