#!/bin/bash
# round 4, second GPU pass: chunks across column groups (looped kernel instantiations).  Targeted parity tests, the A/B on
# view-culled clouds against one group per chunk, the headline (straight-line kernels: must be unchanged), multi-pair routes.
REPO=${GRAFT_REPO_ROOT:-/root/repo}
cd $REPO
O=$REPO/gpurun_out/r04b
mkdir -p $O
export TMPDIR=/tmp
T0=$(date +%s)
el() { echo "[t+$(( $(date +%s) - T0 ))s] $*"; }
echo "== targeted tests"
NIDREG_MARGINS_OUT=$O/parity_margins_targeted.json timeout 900 python -m pytest tests/test_gpu_parity.py -q -m gpu --tb=short -p no:cacheprovider -x \
  -k "chunks_across or deterministic_and_tiling or single_column_special or multi_pair_single_grid or in_library_sharding or nearest_integer or outliers_and_padding or spline_value_gradient or edge_cases or device_resident_cull or odd_bin" > $O/tests_targeted.txt 2>&1; echo "rc=$?"; tail -5 $O/tests_targeted.txt
el "targeted done"
echo "== culled cloud A/B"
timeout 300 python tools/culled_cloud_ab.py 10000000 pinhole_1080p 6.0 | tee $O/culled_cloud_ab.jsonl
timeout 300 python tools/culled_cloud_ab.py 10000000 pinhole_1080p 3.0 | tee -a $O/culled_cloud_ab.jsonl
timeout 300 python tools/culled_cloud_ab.py 10000000 equirect_2k 6.0 | tee -a $O/culled_cloud_ab.jsonl
el "culled done"
echo "== headline sanity"
timeout 120 python tools/make_scene_cache.py /tmp/scene.npz > $O/make_scene.log 2>&1
timeout 120 python tools/run_scene.py /tmp/scene.npz 12 | tee $O/run_scene.json | cut -c1-400
el "headline done"
echo "== multi-pair routes"
python tools/dump_scene_raw.py /tmp/scene.npz /tmp/scene.raw > /dev/null
for K in 2 4 8; do
  OMP_WAIT_POLICY=active timeout 60 tools/omp_pairs.bin 10000000 120 /tmp/scene.raw $K | tee -a $O/omp_pairs.jsonl
done
el "end"
