#!/bin/bash
# round 4, fourth GPU pass: kernel variants on the headline (first-batch prefetch ahead of the prologues; k_entropy shapes),
# the chunk-table choice rule on culled clouds (plain and skewed), the NEAREST kernel with 32-bit LDS cells / five waves.
REPO=${GRAFT_REPO_ROOT:-/root/repo}
cd $REPO
O=$REPO/gpurun_out/r04d
mkdir -p $O
export TMPDIR=/tmp
T0=$(date +%s)
el() { echo "[t+$(( $(date +%s) - T0 ))s] $*"; }
echo "== variants (headline, event-timed kernels, twice for the noise)"
bash tools/run_variants.sh 30 "base:0 nopf:0 ent64:0 ent32:0 base:0 nopf:0 ent64:0 ent32:0" | tee $O/variants.txt
cp gpurun_out/variants.jsonl $O/variants.jsonl
el "variants done"
echo "== targeted tests (prefetch restructure, nearest u32)"
timeout 900 python -m pytest tests/test_gpu_parity.py -q -m gpu --tb=short -p no:cacheprovider -x \
  -k "chunks_across or submit_wait or nearest or deterministic_and_tiling or single_column or multi_pair_single_grid or in_library_sharding or outliers_and_padding or odd_bin or double_records or edge_cases or value_gradient" > $O/tests_targeted.txt 2>&1; echo "rc=$?"; tail -5 $O/tests_targeted.txt
el "targeted done"
echo "== culled cloud A/B (rule: segmented only when it pays)"
timeout 300 python tools/culled_cloud_ab.py 10000000 pinhole_1080p 6.0 | tee $O/culled_cloud_ab.jsonl
timeout 300 python tools/culled_cloud_ab.py 10000000 pinhole_1080p 6.0 skew | tee -a $O/culled_cloud_ab.jsonl
timeout 300 python tools/culled_cloud_ab.py 10000000 equirect_2k 6.0 skew | tee -a $O/culled_cloud_ab.jsonl
NIDREG_SEG_MIN_GAIN=0 timeout 300 python tools/culled_cloud_ab.py 10000000 pinhole_1080p 6.0 skew | tee -a $O/culled_cloud_ab_mingain0.jsonl
el "culled done"
echo "== NEAREST twin"
cd /tmp
timeout 200 rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof_nearest -- python $REPO/tools/run_scene_nearest.py /tmp/scene.npz 60 > $O/nearest.json 2> $O/nearest.err; cut -c1-200 $O/nearest.json
cd $REPO
f=$(find $O/prof_nearest -name "*kernel_stats.csv" | head -1); [ -n "$f" ] && grep "k_nearest\|k_entropy" "$f" | cut -d, -f2-4,6 
el "end"
