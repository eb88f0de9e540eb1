#!/bin/bash
# Collects the measurement set committed under profiles/r01 (run on the GPU box, from the repo root):
#   rm -rf gpurun_out/r1final            # LOCALLY first: gpurun merges into gpurun_out/, so files of an earlier
#                                        # collection (NNN_kernel_stats.csv ...) would otherwise sit beside the new ones
#   gpurun --timeout 1800 -- 'bash scripts/collect_profiles.sh'
# Produces in gpurun_out/r1final: bench.json (default bench incl. cpu_baseline) + tiles.json (its tuned tile / fusion set),
# stats/ (rocprofv3 --kernel-trace --stats of the same tile set; compare its conv_igemm rows / 7 steps with
# stats_bench.json's kernel_ms_per_step), pmc_fetch/ and pmc_write/ (separate --pmc passes, kernel trace only, as the pool
# requires) summarised into hbm_traffic.json by scripts/pmc_summary.py, bench_cfg5.json / bench_cfg2.json (BASELINE.json
# configs 4 and 1).  Copy into profiles/r01 as the *_v3 files; set bench_v3.json's roofline.traffic to
# hbm_traffic.json's conv_igemm_hbm_bytes_per_step / the run's launches per step.  All files of a set come from ONE call.
set -x
cd /tmp; export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
O=gpurun_out/r1final; rm -rf $O; mkdir -p $O
python bench.py --save-tiles $O/tiles.json > $O/bench.json 2> $O/bench.err
rocprofv3 --kernel-trace --stats --output-format csv -d $O/stats -- python3 bench.py --steps 5 --warmup 2 --no-cpu-baseline --load-tiles $O/tiles.json > $O/stats_bench.json 2> $O/stats.err
rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $O/pmc_fetch -- python3 bench.py --steps 3 --warmup 1 --no-cpu-baseline --load-tiles $O/tiles.json > /dev/null 2> $O/pmc_fetch.err
rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d $O/pmc_write -- python3 bench.py --steps 3 --warmup 1 --no-cpu-baseline --load-tiles $O/tiles.json > /dev/null 2> $O/pmc_write.err
python scripts/pmc_summary.py 4 $O/hbm_traffic.json $O/pmc_fetch $O/pmc_write
python bench.py --size 416 --classes 3 --batch 64 --dtype f16 --no-cpu-baseline > $O/bench_cfg5.json 2>/dev/null
python bench.py --batch 1 --dtype f32 --no-cpu-baseline --steps 50 > $O/bench_cfg2.json 2>/dev/null
find $O -name "*.csv" -size +20M -delete
ls -la $O $O/stats/* | head -30
