#!/bin/bash
# timing ablations of conv_halo2_kernel (HALO_ABL bits: 1 no weight loads, 2 no halo pieces, 4 no fragment reads, 8 no barrier)
export YOLO4HIP_LIB=scratch/libyolo4hip_h2abl.so
for a in ${@:-0 1 2 4 8 3 7 15}; do echo "== HALO_ABL=$a"; HALO_ABL=$a timeout 200 python scripts/halo2_bench.py --check 0 --layers 0,1 --tiles 55,58 2>&1 | grep -E "3x3"; done
