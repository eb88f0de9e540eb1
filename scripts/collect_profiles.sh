#!/bin/bash
# Collects the measurement set committed under profiles/r06 (run on the GPU box, from the repo root), ONCE per round:
#   rm -rf gpurun_out/r6set                 # LOCALLY first: gpurun merges into gpurun_out/
#   gpurun --timeout 2700 -- 'Y4_COLLECT_TILES=yolo-v4-tf.keras_amd/yolo4hip/schedules/608_80_32_bf16.json bash scripts/collect_profiles.sh'
# Produces in gpurun_out/r6set (all from ONE call on one box):
#   bench.json + tiles.json        the default bench (two batches in flight; incl. cpu_baseline) and its tuned tile / fusion set
#   bench_single_stream.json       the same tile set with --in-flight 1 (HIP events inside the timed blocks)
#   in_flight_sweep.txt            scripts/two_batches.py 1 / 2 / 3
#   smi_idle.txt, smi_load.txt     rocm-smi clocks / power / temperature / power cap before and (sampled every 0.5 s) during a bench
#   stats/ + stats_bench.json      rocprofv3 --kernel-trace --stats of the single-stream command (conv rows / 7 steps must agree with
#                                  stats_bench.json's kernel_ms_per_step), gaps.json = inter-kernel gaps from that kernel trace;
#   stats2/ + stats2_bench.json    the same of the default (two in flight) command: kernel durations there include the neighbour stream
#   pmc/pass*                      separate --pmc passes of the single-stream command (kernel trace only, program straight after `--`)
#   hbm_traffic.json, mfma_util.json, sq_wait_breakdown.json, pmc_kernels.json
#   bench_cfg5.json / bench_cfg2.json   BASELINE.json configs 5 and 2
set -x
cd /tmp; export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
O=gpurun_out/r6set; rm -rf $O; mkdir -p $O/pmc
SMI="rocm-smi --showclocks --showpower --showtemp --showperflevel --showmaxpower"
$SMI > $O/smi_idle.txt 2>&1
# the schedule of the whole set: $Y4_COLLECT_TILES (a schedule file, e.g. scripts/instep_select.py's) if given, else a fresh autotune on THIS box;
# afterwards tiles.json is what ships as yolo4hip/schedules/608_80_32_bf16.json
if [ -n "$Y4_COLLECT_TILES" ]; then cp "$Y4_COLLECT_TILES" $O/tiles.json; python bench.py --load-tiles $O/tiles.json > $O/bench.json 2> $O/bench.err
else python bench.py --retune --save-tiles $O/tiles.json > $O/bench.json 2> $O/bench.err; fi
# steady single-stream load for ~45 s; the sampler starts once the engine is up and the steps are running
python bench.py --no-cpu-baseline --no-latency --load-tiles $O/tiles.json --in-flight 1 --steps 1500 --blocks 5 > $O/bench_single_stream.json 2> $O/bench_single_stream.err &
BPID=$!
sleep 22
( for i in $(seq 1 24); do echo "== sample $i $(date +%s.%N)"; rocm-smi --showclocks --showpower --showtemp 2>/dev/null | grep -E "sclk|mclk|Power \(W\)|junction"; sleep 0.4; done ) > $O/smi_load.txt 2>&1
wait $BPID
for s in 1 2 3; do python scripts/two_batches.py $s 2>/dev/null | tail -1; done > $O/in_flight_sweep.txt
B1="python3 bench.py --no-cpu-baseline --no-latency --stop-after-conv -1 --load-tiles $O/tiles.json --blocks 1 --in-flight 1"
B2="python3 bench.py --no-cpu-baseline --no-latency --stop-after-conv -1 --load-tiles $O/tiles.json --blocks 1"
rocprofv3 --kernel-trace --stats --output-format csv -d $O/stats -- $B1 --steps 5 --warmup 2 > $O/stats_bench.json 2> $O/stats.err
python scripts/trace_gaps.py $(ls $O/stats/*/*kernel_trace.csv | head -1) $O/gaps.json
rocprofv3 --kernel-trace --stats --output-format csv -d $O/stats2 -- $B2 --steps 6 --warmup 2 > $O/stats2_bench.json 2> $O/stats2.err
i=0
for grp in "FETCH_SIZE" "WRITE_SIZE" \
           "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_VALU_MFMA_BUSY_CYCLES SQ_ACTIVE_INST_VALU SQ_INSTS_VALU" \
           "GRBM_GUI_ACTIVE SQ_WAVES SQ_LDS_BANK_CONFLICT SQ_INSTS_LDS SQ_WAIT_INST_LDS SQ_INSTS_VALU_MFMA_MOPS_BF16 SQ_INSTS_SALU SQ_INSTS_VMEM" \
           "TCC_REQ TCC_HIT TCC_MISS"; do
  i=$((i+1))
  rocprofv3 --pmc $grp --kernel-trace --output-format csv -d $O/pmc/pass$i -- $B1 --steps 3 --warmup 1 > /dev/null 2> $O/pmc/pass$i.err
  find $O/pmc/pass$i -name "*kernel_trace.csv" -delete; find $O/pmc/pass$i -name "*agent_info.csv" -delete
done
python scripts/pmc_summary.py 4 $O/hbm_traffic.json $O/pmc/pass1 $O/pmc/pass2
python scripts/pmc_kernels.py $O/pmc_kernels.json $O/pmc/pass3 $O/pmc/pass4 $O/pmc/pass5 > $O/pmc_kernels.txt
python scripts/mfma_util.py $O/pmc_kernels.json - $O/mfma_util.json $O/sq_wait_breakdown.json > $O/mfma_util.txt
python bench.py --size 416 --classes 3 --batch 64 --dtype f16 --no-cpu-baseline --no-latency > $O/bench_cfg5.json 2>/dev/null
python bench.py --batch 1 --dtype f32 --no-cpu-baseline --no-latency --steps 50 > $O/bench_cfg2.json 2>/dev/null
# round 5: fp16 at the headline shape, the halo tiles against the implicit-GEMM tiles layer by layer, the determinism hunt
python bench.py --dtype f16 --no-cpu-baseline --no-latency > $O/bench_608_80_32_f16.json 2>/dev/null
python bench.py --dtype f16 --no-cpu-baseline --no-latency --in-flight 1 > $O/bench_608_80_32_f16_single_stream.json 2>/dev/null
python scripts/halo_bench.py > $O/halo_bench_bf16.txt 2>&1
python scripts/halo_bench.py --dtype f16 > $O/halo_bench_f16.txt 2>&1
timeout 600 python scripts/determinism_hunt.py --sweep --iters 500 > $O/determinism_hunt.txt 2>&1
# round 6: the halo2 tiles (conv_halo2_kernel.h) layer by layer, hot and L2-cold; what the matrix pipes sustain by operand data; the
# race screens of the halo2 tiles (one conv under a second stream's load; the shipped schedule inside the engine)
python scripts/halo2_bench.py > $O/halo2_bench_bf16.txt 2>&1
python scripts/halo2_bench.py --dtype f16 --check 0 > $O/halo2_bench_f16.txt 2>&1
python scripts/halo2_bench.py --cold 64 --check 0 > $O/halo2_bench_bf16_cold.txt 2>&1
for f in randn zero; do echo "== fill $f"; python scripts/halo2_bench.py --check 0 --layers 0,1 --tiles 55,59,61 --fill $f 2>&1 | grep 3x3; done > $O/halo2_bench_fill.txt 2>&1
if [ -x scratch/ubench/mfma_power ]; then for c in 256 128 64; do scratch/ubench/mfma_power $c 60; done > $O/mfma_power.txt 2>&1; fi
for dt in bf16 f16; do for act in 1 2; do echo "== $dt act $act"; python scripts/h2_stress.py --reps 200 --dtype $dt --act $act 2>&1 | grep -v amdgpu.ids; done; done > $O/h2_stress.txt 2>&1
for dt in bf16 f16; do python scripts/h2_det.py --dtype $dt --forwards 200 2>&1 | grep -v amdgpu.ids | tail -3; done > $O/h2_det.txt 2>&1
# one image: the kernel timeline of a step on the shipped latency schedules (rocprofv3 --kernel-trace), bf16 and fp32
for dt in bf16 f32; do
  rocprofv3 --kernel-trace --output-format csv -d $O/b1_$dt -- python3 bench.py --batch 1 --dtype $dt --no-cpu-baseline --no-latency --in-flight 1 --steps 40 --warmup 5 --blocks 1 > $O/b1_$dt.json 2> $O/b1_$dt.err
  python scripts/step_timeline.py $(ls $O/b1_$dt/*/*kernel_trace.csv | head -1) > $O/b1_${dt}_timeline.txt 2>&1
  rm -rf $O/b1_$dt
done
find $O -name "*.csv" -size +20M -delete
find $O -name "*agent_info.csv" -delete
ls -la $O $O/stats/* | head -40
