#!/bin/bash
# N tuning runs of the headline shape with the halo2 tiles on offer, majority vote per launch
set -u
out=${OUT:-gpurun_out/r6/sched}; mkdir -p $out
N=${1:-5}; name=${2:-608_80_32_bf16}; shift 2
export YOLO4HIP_HALO2=1
files=""
for i in $(seq 1 $N); do
  python bench.py --no-cpu-baseline --no-latency --retune --steps 5 --blocks 1 --save-tiles $out/${name}_run$i.json "$@" > $out/${name}_run$i.bench.json 2>/dev/null
  files="$files $out/${name}_run$i.json"
done
python scripts/vote_schedule.py $out/$name.json $files > $out/$name.vote.txt
tail -3 $out/$name.vote.txt
python - <<PY
import json
t=json.load(open("$out/$name.json"))["tiles"]
print("halo2 picks:", [(i,x) for i,x in enumerate(t) if 55 <= abs(x)%100 <= 70 and abs(x)%1000 < 100])
PY
