#!/bin/bash
# timing ablations of conv_halo_kernel (HALO_ABL bits: 1 no weight loads in the loop, 2 no halo-tile loads, 4 no barrier).
# Needs a variant library built with the switches compiled in:
#   bash scripts/build_variant2.sh haloabl "-DHALO_ABLATIONS=1" conv_halo_bf16 conv_halo_f16   ->  scratch/libyolo4hip_haloabl.so
# and YOLO4HIP_LIB=scratch/libyolo4hip_haloabl.so in the environment; the regular build ignores HALO_ABL.
mkdir -p gpurun_out/r5
for a in ${@:-0 16 1 3 4}; do echo "== HALO_ABL=$a"; HALO_ABL=$a timeout 120 python scripts/halo_bench.py 2>&1 | grep -E "256->512|512->1024|256->256|512->512"; done
