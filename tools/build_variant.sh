#!/bin/bash
# A/B build of the library with extra macros, beside the production one:  bash tools/build_variant.sh <name> "<-D flags>" [sources...]
#   -> cdnet_amd/libcdnet_hip_<name>.so (select with CDNET_LIB_PATH); only the listed sources (default wgrad.hip conv32ws.hip) are
#   recompiled, the other objects come from the production build (python -m cdnet_amd.csrc.build first).
NAME=$1; FLAGS=$2; shift 2
SRCS=${@:-wgrad.hip conv32ws.hip}
ROOT=$(cd $(dirname $0)/.. && pwd)
OBJ=$ROOT/cdnet_amd/csrc/build
TMP=/tmp/cdnet_variant_$NAME
mkdir -p $TMP
OBJS=""
for o in $OBJ/*.o; do
  b=$(basename $o .o)
  if echo " $SRCS " | grep -q " $b.hip "; then
    hipcc --offload-arch=gfx950 -O3 -std=c++20 -fPIC -ffp-contract=off -Wno-unused-function -Wno-unused-result $FLAGS -I$ROOT/include -c $ROOT/cdnet_amd/csrc/$b.hip -o $TMP/$b.o &
    OBJS="$OBJS $TMP/$b.o"
  else
    OBJS="$OBJS $o"
  fi
done
wait
hipcc --offload-arch=gfx950 -shared -fPIC -o $ROOT/cdnet_amd/libcdnet_hip_$NAME.so $OBJS && echo built $ROOT/cdnet_amd/libcdnet_hip_$NAME.so
