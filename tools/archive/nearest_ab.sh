cd $GRAFT_REPO_ROOT
for cam in equirect_2k:10000000 fisheye_1080p:5000000 omnidir_2k:10000000 pinhole_1080p:10000000; do
  c=${cam%%:*}; n=${cam##*:}
  f=/tmp/scene_${c}_${n}.npz
  [ -f $f ] || python tools/make_scene_cache.py $f $c $n 20250530 > /dev/null 2>&1
  for mode in fast exact; do
    if [ $mode = exact ]; then export NIDREG_NEAREST_EXACT=1; else unset NIDREG_NEAREST_EXACT; fi
    echo "$c $mode $(python tools/run_scene_nearest.py $f 100 256 2>/dev/null | tail -1 | cut -c1-140)"
  done
done
