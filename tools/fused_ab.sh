# A/B of the fused single-launch evaluation (csrc/nid_fused.hpp) against the three-kernel route on the same box, same process order:
# usage fused_ab.sh <out.jsonl>   (NIDREG_FUSED=0 / 1 alternating, three rounds; 16 bins: the reference's default)
cd $GRAFT_REPO_ROOT
out=${1:-gpurun_out/fused_ab.jsonl}
: > $out
for round in $(seq 1 ${ROUNDS:-3}); do
  for f in 0 1; do
    for spec in "16 100000 0,64,128,256" "16 300000 0,128,256" "16 1000000 0,256,512" "32 100000 0"; do
      set -- $spec
      NIDREG_FUSED=$f python tools/small_cloud_sweep.py $1 $2 $3 2>/dev/null | tail -1 | sed "s/^{/{\"fused_env\": $f, \"round\": $round, /" >> $out
    done
  done
done
