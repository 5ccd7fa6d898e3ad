# the concurrency / sharding tests N times over (a race shows up as one failing run in many): usage repeat_concurrency.sh [N]
cd $GRAFT_REPO_ROOT
fails=0
for i in $(seq 1 ${1:-10}); do
  timeout 600 python -m pytest tests/test_concurrent_callers.py tests/test_sharded_concurrent.py -q -m gpu -p no:cacheprovider -x > /tmp/repeat_$i.txt 2>&1 || { fails=$((fails+1)); tail -20 /tmp/repeat_$i.txt; }
  tail -1 /tmp/repeat_$i.txt
done
echo "failed runs: $fails of ${1:-10}"
