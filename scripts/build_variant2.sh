#!/bin/bash
# usage: build_variant2.sh <name> "<extra flags>" unit [unit ...]  -> scratch/libyolo4hip_<name>.so (several units recompiled)
set -e
cd "$(dirname "$0")/../yolo-v4-tf.keras_amd/csrc"
NAME=$1; EXTRA=$2; shift 2
mkdir -p ../../scratch/obj
pids=()
for U in "$@"; do
  hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -ffp-contract=off -Wno-unused-function -Wno-unused-variable $EXTRA -c $U.hip -o ../../scratch/obj/${U}_$NAME.o &
  pids+=($!)
done
for p in "${pids[@]}"; do wait $p; done
OBJS=""
for f in build/*.o; do
  b=$(basename $f .o); use=$f
  for U in "$@"; do if [ "$b" == "$U" ]; then use=../../scratch/obj/${U}_$NAME.o; fi; done
  OBJS="$OBJS $use"
done
hipcc --offload-arch=gfx950 -shared -fPIC -o ../../scratch/libyolo4hip_$NAME.so $OBJS
echo "built scratch/libyolo4hip_$NAME.so"
