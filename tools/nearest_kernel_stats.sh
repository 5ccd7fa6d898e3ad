# rocprofv3 kernel durations of the NEAREST twin (k_nearest_hist, k_entropy) per camera model, fast tier on / off
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
for cam in pinhole_1080p:10000000 equirect_2k:10000000 omnidir_2k:10000000 fisheye_1080p:5000000; do
  c=${cam%%:*}; n=${cam##*:}
  f=/tmp/scene_${c}_${n}.npz
  [ -f $f ] || python tools/make_scene_cache.py $f $c $n 20250530 > /dev/null 2>&1
  for mode in fast exact; do
    if [ $mode = exact ]; then export NIDREG_NEAREST_EXACT=1; else unset NIDREG_NEAREST_EXACT; fi
    d=/tmp/nk_${c}_$mode
    rm -rf $d
    (cd /tmp && rocprofv3 --kernel-trace --stats --output-format csv -d $d -- python $GRAFT_REPO_ROOT/tools/run_scene_nearest.py $f 60 256 > /dev/null 2>&1)
    F=$(find $d -name "*kernel_stats.csv" | head -1)
    echo "$c $n $mode $(python - "$F" <<'PY'
import csv, re, sys
out = {}
for r in csv.DictReader(open(sys.argv[1])):
    m = re.search(r"nidreg::(k_nearest_hist|k_entropy)", r["Name"])
    if m:
        out[m.group(1)] = f"{float(r['AverageNs']) / 1e3:.2f} us x {r['Calls']}"
print(out)
PY
)"
  done
done
