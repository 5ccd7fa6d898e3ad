#!/bin/bash
# One GPU pass of a round: `gpurun -- bash tools/round_pass.sh <tag> <step> [<step> ...]` (replaces the per-pass scripts of
# rounds 3-4, tools/rounds/ -- their measurements live on in profiles/archive/r03*_*, r04*_*).  Everything lands in gpurun_out/<tag>/;
# what is to be judged is copied to profiles/ by hand afterwards.  Steps (run in the order given):
#   tests[=<pytest -k expression>]      the -m gpu suite (or a subset) with the parity-margin recorder
#   stats=<camera>:<points>[:<bins>]    rocprofv3 --kernel-trace --stats around tools/run_scene.py on a cached scene
#                                       -> <camera>_<points>_kernel_stats.{csv,json} (json stamped with the kernel build)
#   pmc=<camera>:<points>[:<passes>]    PMC counters in separate rocprofv3 passes (tools/profile_pmc.sh) -> <camera>_<points>_traffic.json,
#                                       <camera>_<points>_pmc_summary.txt        (passes: "fetch write sq1 sq2 sq3 tcc tcp", + separated)
#   run=<camera>:<points>[:<bins>]      tools/run_scene.py alone (event times, wall time) -> <camera>_<points>_run.json
#   variants=<camera>:<points>:<v1,v2>  variants/libnidreg_<v>.so (tools/build_variants.sh) on that scene, same box -> variants.jsonl
#   benchstats                          rocprofv3 --kernel-trace --stats around the headline bench command -> kernel_stats.json
#   bench[=<bench.py arguments>]        python bench.py ... -> bench_line.json
#   shardcost[=<bins>]                  tools/shard_cost.py (one-GPU protocol measurement) -> shard_cost_b<bins>.json
#   sh=<command>                        anything else, logged to sh_<n>.txt
REPO=${GRAFT_REPO_ROOT:-/root/repo}
cd $REPO
TAG=${1:?tag}; shift
O=$REPO/gpurun_out/$TAG
mkdir -p $O
export TMPDIR=/tmp
T0=$(date +%s)
el() { echo "[t+$(( $(date +%s) - T0 ))s] $*"; }
dims() { python - "$1" <<'PY'
import sys
sys.path.insert(0, ".")
from direct_visual_lidar_calibration_amd import synth
c = synth.CONFIG_CAMERAS[sys.argv[1]]
print(c[3], c[4])
PY
}
scene() {  # camera points -> path of the cached scene (seed = bench.py's config legs)
  local f=/tmp/scene_$1_$2.npz
  [ -f $f ] || python tools/make_scene_cache.py $f $1 $2 $((20250523 + 7)) > $O/make_scene_$1_$2.log 2>&1
  echo $f
}
N=0
for STEP in "$@"; do
  N=$((N + 1))
  KIND=${STEP%%=*}; ARG=""; [ "$KIND" != "$STEP" ] && ARG=${STEP#*=}
  IFS=: read -r A1 A2 A3 <<< "$ARG"
  echo "== [$N] $STEP"
  case $KIND in
    tests)
      if [ -n "$ARG" ]; then
        NIDREG_MARGINS_OUT=$O/parity_margins_$N.json timeout 1500 python -m pytest tests -q -m gpu --tb=short -p no:cacheprovider -k "$ARG" > $O/tests_$N.txt 2>&1
      else
        NIDREG_MARGINS_OUT=$O/parity_margins.json timeout 1700 python -m pytest tests -q -m gpu --tb=short -p no:cacheprovider > $O/tests_gpu.txt 2>&1
      fi
      echo "rc=$?"; tail -n 6 $O/tests_*.txt | tail -n 8 ;;
    stats)
      S=$(scene $A1 $A2); B=${A3:-256}; read W H <<< "$(dims $A1)"
      (cd /tmp && timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/trace_$N -- python $REPO/tools/run_scene.py $S 10 fp64 $B > $O/${A1}_${A2}_run.json 2> $O/${A1}_${A2}_run_err.txt); echo "rc=$?"
      F=$(find $O/trace_$N -name "*kernel_stats.csv" | head -1)
      if [ -n "$F" ]; then
        grep -E "Name|nidreg" $F > $O/${A1}_${A2}_kernel_stats.csv
        python tools/kernel_stats_json.py $F $O/${A1}_${A2}_kernel_stats.json $A2 $W $H $B fp64 "rocprofv3 --kernel-trace --stats -- python tools/run_scene.py (scene $A1, $A2 points, seed 20250530)" $A1
      fi
      rm -rf $O/trace_$N; tail -c 600 $O/${A1}_${A2}_run.json ;;
    pmc)
      S=$(scene $A1 $A2); read W H <<< "$(dims $A1)"
      PASSES=${A3:-fetch+write+sq1+sq2}
      SCENE_NPZ=$S PMC_PASSES="${PASSES//+/ }" timeout 900 bash tools/profile_pmc.sh ${TAG}_${A1} > $O/${A1}_${A2}_pmc.log 2>&1; echo "rc=$?"
      cp gpurun_out/pmc_${TAG}_${A1}/summary.txt $O/${A1}_${A2}_pmc_summary.txt
      python tools/traffic_from_pmc.py gpurun_out/pmc_${TAG}_${A1} $O/${A1}_${A2}_traffic.json $A2 $W $H 256 fp64 $A1 > /dev/null 2>&1; echo "traffic rc=$?"
      rm -rf gpurun_out/pmc_${TAG}_${A1} ;;
    run)
      S=$(scene $A1 $A2)
      timeout 300 python tools/run_scene.py $S 20 fp64 ${A3:-256} > $O/${A1}_${A2}_run.json 2> $O/${A1}_${A2}_run_err.txt; echo "rc=$?"; tail -c 700 $O/${A1}_${A2}_run.json ;;
    variants)
      S=$(scene $A1 $A2)
      for V in ${A3//,/ }; do
        NIDREG_LIB=$REPO/variants/libnidreg_$V.so timeout 200 python tools/run_scene.py $S 20 fp64 256 2>&1 | tail -1 | sed "s/^{/{\"variant\": \"$V\", \"camera\": \"$A1\", \"points\": $A2, /" >> $O/variants.jsonl
      done
      python - $O/variants.jsonl <<'PY'
import json, sys
for l in open(sys.argv[1]):
    try:
        d = json.loads(l)
    except Exception:
        print("BAD", l[:200]); continue
    k = d["kernel_ms"]
    print(f"{d['camera']:14s} {d['variant']:16s} batch={d.get('wall_batch_ms')} hist={k.get('hist')} ent={k.get('entropy')} grad={k.get('grad')} cost={d['last_cost']!r} g0={d['last_grad'][0]!r}")
PY
      ;;
    benchstats)
      (cd /tmp && timeout 200 rocprofv3 --kernel-trace --stats --output-format csv -d $O/trace_$N -- python $REPO/bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-config-legs > $O/bench_trace.log 2>&1); echo "rc=$?"
      F=$(find $O/trace_$N -name "*kernel_stats.csv" | head -1)
      if [ -n "$F" ]; then
        grep -E "Name|nidreg" $F > $O/bench_kernel_stats.csv
        python tools/kernel_stats_json.py $F $O/kernel_stats.json 10000000 1920 1080 256 fp64 "rocprofv3 --kernel-trace --stats -- python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-config-legs"
      fi
      rm -rf $O/trace_$N ;;
    bench)
      timeout 1200 python bench.py $ARG > $O/bench_line_$N.json 2> $O/bench_err_$N.txt; echo "rc=$?"; cut -c1-600 $O/bench_line_$N.json ;;
    shardcost)
      timeout 600 python tools/shard_cost.py ${ARG:-256} > $O/shard_cost_b${ARG:-256}.json 2> $O/shard_cost_err.txt; echo "rc=$?"; cat $O/shard_cost_b${ARG:-256}.json | head -c 1500 ;;
    sh)
      timeout 1500 bash -c "$ARG" > $O/sh_$N.txt 2>&1; echo "rc=$?"; tail -n 15 $O/sh_$N.txt ;;
    *) echo "unknown step $STEP" ;;
  esac
  el "$STEP done"
done
