#!/bin/bash
# Collects the measurement set committed under profiles/r02 (run on the GPU box, from the repo root):
#   rm -rf gpurun_out/r2set                 # LOCALLY first: gpurun merges into gpurun_out/
#   gpurun --timeout 2400 -- 'bash scripts/collect_profiles.sh'
# Produces in gpurun_out/r2set (all from ONE call on one box):
#   bench.json + tiles.json   default bench (incl. cpu_baseline) and its tuned tile / fusion set
#   stats/ + stats_bench.json rocprofv3 --kernel-trace --stats of the same tile set (conv rows / 7 steps must agree with
#                             stats_bench.json's kernel_ms_per_step), gaps.json = inter-kernel gaps from that kernel trace
#   pmc/pass*                 separate --pmc passes (kernel trace only, program straight after `--`, as the pool requires):
#                             FETCH_SIZE | WRITE_SIZE | SQ wave-cycle breakdown + MFMA busy | GRBM_GUI_ACTIVE + LDS | L2
#   hbm_traffic.json (scripts/pmc_summary.py), mfma_util.json + sq_wait_breakdown.json (scripts/mfma_util.py)
#   bench_cfg5.json / bench_cfg2.json   BASELINE.json configs 5 and 2
set -x
cd /tmp; export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
O=gpurun_out/r2set; rm -rf $O; mkdir -p $O/pmc
python bench.py --save-tiles $O/tiles.json > $O/bench.json 2> $O/bench.err
B="python3 bench.py --no-cpu-baseline --load-tiles $O/tiles.json --blocks 1"
rocprofv3 --kernel-trace --stats --output-format csv -d $O/stats -- $B --steps 5 --warmup 2 > $O/stats_bench.json 2> $O/stats.err
python scripts/trace_gaps.py $(ls $O/stats/*/*kernel_trace.csv | head -1) $O/gaps.json
i=0
for grp in "FETCH_SIZE" "WRITE_SIZE" \
           "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_VALU_MFMA_BUSY_CYCLES SQ_ACTIVE_INST_VALU SQ_INSTS_VALU" \
           "GRBM_GUI_ACTIVE SQ_WAVES SQ_LDS_BANK_CONFLICT SQ_INSTS_LDS SQ_WAIT_INST_LDS SQ_INSTS_VALU_MFMA_MOPS_BF16 SQ_INSTS_SALU SQ_INSTS_VMEM" \
           "TCC_REQ TCC_HIT TCC_MISS"; do
  i=$((i+1))
  rocprofv3 --pmc $grp --kernel-trace --output-format csv -d $O/pmc/pass$i -- $B --steps 3 --warmup 1 > /dev/null 2> $O/pmc/pass$i.err
  find $O/pmc/pass$i -name "*kernel_trace.csv" -delete; find $O/pmc/pass$i -name "*agent_info.csv" -delete
done
python scripts/pmc_summary.py 4 $O/hbm_traffic.json $O/pmc/pass1 $O/pmc/pass2
python scripts/pmc_kernels.py $O/pmc_kernels.json $O/pmc/pass3 $O/pmc/pass4 $O/pmc/pass5 > $O/pmc_kernels.txt
python scripts/mfma_util.py $O/pmc_kernels.json - $O/mfma_util.json $O/sq_wait_breakdown.json > $O/mfma_util.txt
python bench.py --size 416 --classes 3 --batch 64 --dtype f16 --no-cpu-baseline > $O/bench_cfg5.json 2>/dev/null
python bench.py --batch 1 --dtype f32 --no-cpu-baseline --steps 50 > $O/bench_cfg2.json 2>/dev/null
find $O -name "*.csv" -size +20M -delete
find $O -name "*agent_info.csv" -delete
ls -la $O $O/stats/* | head -40
