#!/bin/bash
# A variant of the library for a same-box A/B (scripts/ab_lib.sh): one launch unit recompiled with extra defines, the
# other objects taken from build/hip/.   usage: scripts/build_variant_lib.sh <name> <unit: row4|row5|attn|panel|...> <defines...>
set -eu
cd "$(dirname "$0")/.."
NAME=$1; UNIT=$2; shift 2
mkdir -p ab_libs build/variant
case $UNIT in
  row[0-9]) SRC=open_provence_amd/csrc/op_launch_row.hip; PART="-DOPL_ROW_PART=${UNIT#row}"; OBJ=op_launch_$UNIT ;;
  attn|panel|layer32) SRC=open_provence_amd/csrc/op_launch_$UNIT.hip; PART=""; OBJ=op_launch_$UNIT ;;
  api) SRC=open_provence_amd/csrc/op_api.hip; PART=""; OBJ=op_api ;;
  *) echo "unknown unit $UNIT"; exit 1 ;;
esac
hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -Wno-unused-result -fno-slp-vectorize $PART "$@" -c $SRC -o build/variant/${NAME}_$OBJ.o
OBJS=""
for o in build/hip/*.o; do
  b=$(basename $o .o)
  if [ "$b" = "$OBJ" ]; then OBJS="$OBJS build/variant/${NAME}_$OBJ.o"; else OBJS="$OBJS $o"; fi
done
hipcc --offload-arch=gfx950 -shared -fPIC -o ab_libs/$NAME.so $OBJS
ls -la ab_libs/$NAME.so
