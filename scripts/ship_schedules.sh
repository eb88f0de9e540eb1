#!/bin/bash
# Re-tunes every schedule that ships with the package on the GPU box (round 5: the halo tiles 51-54 exist now).
#   headline shapes (batch 32 / 64): N tuning runs of bench.py --retune, majority vote per launch (scripts/vote_schedule.py)
#   the others: one Engine.ensure_schedule run each (scripts/make_schedules.py, Y4_RETUNE=1)
# Output: gpurun_out/r5/sched/<shape>.json -- copy into yolo-v4-tf.keras_amd/yolo4hip/schedules/ to ship.
set -u
out=gpurun_out/r5/sched; mkdir -p $out
N=${1:-5}
vote() {   # name, bench args...
  name=$1; shift
  files=""
  for i in $(seq 1 $N); do
    python bench.py --no-cpu-baseline --no-latency --retune --steps 5 --blocks 1 --save-tiles $out/${name}_run$i.json "$@" > $out/${name}_run$i.bench.json 2>/dev/null
    files="$files $out/${name}_run$i.json"
  done
  python scripts/vote_schedule.py $out/$name.json $files > $out/$name.vote.txt
  tail -3 $out/$name.vote.txt
}
vote 608_80_32_bf16 --dtype bf16
vote 608_80_32_f16 --dtype f16
vote 416_3_64_f16 --size 416 --classes 3 --batch 64 --dtype f16
Y4_RETUNE=1 python scripts/make_schedules.py $out 416_80_32_bf16 416_80_1_bf16 608_80_1_bf16 608_80_1_f32 416_80_1_f32 416_80_32_f32 608_80_32_f32 2>&1 | tail -8
ls $out/*.json | grep -v run | head -20
