#!/bin/bash
# PMC counters for the NID kernels via the torch-free driver (separate rocprofv3 passes; only
# --kernel-trace next to --pmc).  Usage: profile_pmc.sh <tag> [run_scene args after the npz...]
TAG=${1:-r1}
shift
ARGS="$@"
REPO=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$REPO/gpurun_out/pmc_$TAG
mkdir -p $OUT
SCENE=${SCENE_NPZ:-/tmp/scene.npz}   # SCENE_NPZ / SCENE_CAMERA / SCENE_POINTS / SCENE_SEED: another cached scene (tools/round_pass.sh pmc:<camera>:<points>)
[ -f $SCENE ] || python $REPO/tools/make_scene_cache.py $SCENE ${SCENE_CAMERA:-pinhole_1080p} ${SCENE_POINTS:-10000000} ${SCENE_SEED:-20250525} > $OUT/make_scene.log 2>&1
DRIVER=${PMC_DRIVER:-run_scene.py}   # PMC_DRIVER=run_scene_nearest.py: the NEAREST twin
export RUN_SCENE_QUICK=1
cd /tmp && export TMPDIR=/tmp
PASSES=${PMC_PASSES:-fetch write sq1 sq2 sq3 tcc tcp}  # PMC_PASSES="fetch write sq1": only what traffic_from_pmc.py / bench.py read
run() { # name, counters...
  NAME=$1; shift
  case " $PASSES " in *" $NAME "*) ;; *) return;; esac
  rocprofv3 --pmc "$@" --kernel-trace --output-format csv -d $OUT/$NAME -- python $REPO/tools/$DRIVER $SCENE 6 $ARGS > $OUT/$NAME.log 2>&1
  echo "$NAME rc=$?"
}
run fetch FETCH_SIZE
run write WRITE_SIZE
run sq1 SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_SALU SQ_ACTIVE_INST_VALU
run sq2 SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_WAIT_INST_LDS SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_LDS_ADDR_CONFLICT
run sq3 SQ_INST_CYCLES_VMEM SQ_ACTIVE_INST_VMEM SQ_LDS_ATOMIC_RETURN SQ_LDS_UNALIGNED_STALL SQ_INSTS_LDS GRBM_GUI_ACTIVE
run tcc TCC_HIT_sum TCC_MISS_sum
run tcp TCP_TOTAL_CACHE_ACCESSES_sum TCP_TCC_READ_REQ_sum
cd $REPO && python tools/pmc_summary.py $OUT > $OUT/summary.txt 2>&1; cat $OUT/summary.txt | head -60
