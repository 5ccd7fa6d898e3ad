#!/bin/bash
# round 3, GPU pass g: multi-pair single grid with the XCD-aware chunk order against the concatenated order and per-pair launches
REPO=${GRAFT_REPO_ROOT:-/root/repo}
cd $REPO
O=$REPO/gpurun_out/r03g
mkdir -p $O
export TMPDIR=/tmp
timeout 300 python tools/make_scene_cache.py /tmp/scene.npz > $O/make_scene.log 2>&1
python tools/dump_scene_raw.py /tmp/scene.npz /tmp/scene.raw > /dev/null
export OMP_WAIT_POLICY=active
echo "xcd-aware order";   timeout 600 tools/omp_pairs.bin 10000000 120 /tmp/scene.raw | tee -a $O/omp_pairs.jsonl
echo "concatenated order"; NIDREG_GROUP_FLAT_ORDER=1 timeout 600 tools/omp_pairs.bin 10000000 120 /tmp/scene.raw | tee -a $O/omp_pairs.jsonl
echo "per-pair launches";  NIDREG_NO_MULTI_GRID=1 timeout 600 tools/omp_pairs.bin 10000000 120 /tmp/scene.raw | tee -a $O/omp_pairs.jsonl
echo "== tests"
timeout 900 python -m pytest tests/test_gpu_parity.py tests/test_calibration.py -q -m gpu --tb=short -p no:cacheprovider -k "multi or calibrat or thread" > $O/tests.txt 2>&1; echo "rc=$?"; tail -4 $O/tests.txt
