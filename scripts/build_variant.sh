#!/bin/bash
# usage: build_variant.sh <name> <unit> "<extra hipcc flags>"  -> scratch/libyolo4hip_<name>.so
# Recompiles ONE translation unit with extra -D flags and links it with the objects of the regular build (build.py first).
# Load it with YOLO4HIP_LIB=scratch/libyolo4hip_<name>.so (kernel experiments; same ABI).
set -e
cd "$(dirname "$0")/../yolo-v4-tf.keras_amd/csrc"
NAME=$1; UNIT=$2; EXTRA=$3
mkdir -p ../../scratch/obj
hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -ffp-contract=off -Wno-unused-function -Wno-unused-variable $EXTRA -c $UNIT.hip -o ../../scratch/obj/${UNIT}_$NAME.o
OBJS=""
for f in build/*.o; do
  b=$(basename $f .o)
  if [ "$b" == "$UNIT" ]; then OBJS="$OBJS ../../scratch/obj/${UNIT}_$NAME.o"; else OBJS="$OBJS $f"; fi
done
hipcc --offload-arch=gfx950 -shared -fPIC -o ../../scratch/libyolo4hip_$NAME.so $OBJS
echo "built scratch/libyolo4hip_$NAME.so"
