#!/bin/bash
# Build experiment variants of libnidreg.so (compile-time macros) into variants/ so that one gpurun call
# can time them all on the same cached scene (NIDREG_LIB selects the library).
# Usage: build_variants.sh name1="-DFLAG ..." name2="..."
REPO=$(cd "$(dirname "$0")/.." && pwd)
mkdir -p $REPO/variants
for spec in "$@"; do
  name=${spec%%=*}; flags=${spec#*=}
  tmp=$(mktemp -d /tmp/nidvar.XXXX)
  mkdir -p $tmp/a/b/csrc $tmp/a/include
  cp $REPO/direct_visual_lidar_calibration_amd/csrc/*.h* $REPO/direct_visual_lidar_calibration_amd/csrc/Makefile $tmp/a/b/csrc/
  cp -r $REPO/include/* $tmp/a/include/
  ( make -j6 -C $tmp/a/b/csrc EXTRA="$flags" > $tmp/build.log 2>&1 && cp $tmp/a/b/csrc/libnidreg.so $REPO/variants/libnidreg_$name.so && echo "built $name" || { echo "FAILED $name"; grep -E "error" $tmp/build.log | head -5; } ) &
done
wait
ls -la $REPO/variants
