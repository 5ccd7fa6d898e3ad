# A/B of two library builds on small / medium clouds: usage small_ab.sh <variantA> <variantB>   (variants/libnidreg_<v>.so)
cd $GRAFT_REPO_ROOT
for v in "$@" "$@" "$@"; do
  for spec in "16 100000" "16 1000000" "256 4096" "256 100000" "256 1000000"; do
    set -- $spec
    echo "$v bins=$1 points=$2 $(NIDREG_LIB=$PWD/variants/libnidreg_$v.so python tools/small_cloud_sweep.py $1 $2 0 2>/dev/null | tail -1 | cut -c1-200)"
  done
done
