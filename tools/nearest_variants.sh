# wall time of the NEAREST twin per evaluation for library variants (variants/libnidreg_<v>.so) on the cached config scenes
# usage: nearest_variants.sh "camera:points ..." v1 v2 ...
cd $GRAFT_REPO_ROOT
cams=$1; shift
for round in 1 2; do
for cam in $cams; do
  c=${cam%%:*}; n=${cam##*:}
  f=/tmp/scene_${c}_${n}.npz
  [ -f $f ] || python tools/make_scene_cache.py $f $c $n 20250530 > /dev/null 2>&1
  for v in "$@"; do
    echo "$c $v $(NIDREG_LIB=$PWD/variants/libnidreg_$v.so python tools/run_scene_nearest.py $f 100 256 2>/dev/null | tail -1 | cut -c1-120)"
  done
done
done
