#!/bin/bash
# round 4, seventh GPU pass: NOCLAMP variant, omp_pairs with the caller's empty parallel region, shard protocol cost
REPO=${GRAFT_REPO_ROOT:-/root/repo}
cd $REPO
O=$REPO/gpurun_out/r04g
mkdir -p $O
export TMPDIR=/tmp
T0=$(date +%s)
el() { echo "[t+$(( $(date +%s) - T0 ))s] $*"; }
echo "== variants"
bash tools/run_variants.sh 30 "base:0 noclamp:0 base:0 noclamp:0 base:0 noclamp:0" | tee $O/variants.txt
el "variants done"
echo "== omp_pairs"
python tools/dump_scene_raw.py /tmp/scene.npz /tmp/scene.raw > /dev/null
for K in 2 4 8; do
  OMP_WAIT_POLICY=active timeout 60 tools/omp_pairs.bin 10000000 120 /tmp/scene.raw $K | tee -a $O/omp_pairs.jsonl
  NIDREG_COHORT=1 OMP_WAIT_POLICY=active timeout 60 tools/omp_pairs.bin 10000000 120 /tmp/scene.raw $K | tee -a $O/omp_pairs.jsonl
done
el "omp done"
echo "== shard protocol cost (one GPU listed n times)"
timeout 300 python tools/shard_cost.py 256 | tee $O/shard_protocol_cost_b256.json | cut -c1-1500
el "end"
