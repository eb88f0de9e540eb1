#!/bin/bash
# timing ablations of conv_halo_kernel (HALO_ABL bits: 1 no weight loads in the loop, 2 no halo-tile loads, 4 no barrier, 16 no sched_barrier pins)
mkdir -p gpurun_out/r5
for a in ${@:-0 16 1 3 4}; do echo "== HALO_ABL=$a"; HALO_ABL=$a timeout 120 python scripts/halo_bench.py 2>&1 | grep -E "256->512|512->1024|256->256|512->512"; done
