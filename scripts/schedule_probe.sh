mkdir -p gpurun_out/sched
python bench.py --size 416 --classes 3 --batch 64 --dtype f16 --no-cpu-baseline --retune --save-tiles gpurun_out/sched/416_3_64_f16.json > gpurun_out/sched/c5_tune.json 2>/dev/null
python bench.py --size 416 --classes 3 --batch 64 --dtype f16 --no-cpu-baseline --load-tiles gpurun_out/sched/416_3_64_f16.json > gpurun_out/sched/c5_load.json 2>/dev/null
python bench.py --size 416 --classes 3 --batch 64 --dtype f16 --no-cpu-baseline --retune > gpurun_out/sched/c5_tune2.json 2>/dev/null
python bench.py --retune --no-cpu-baseline > gpurun_out/sched/h_tune.json 2>/dev/null
python bench.py --no-cpu-baseline > gpurun_out/sched/h_ship.json 2>/dev/null
python bench.py --retune --no-cpu-baseline > gpurun_out/sched/h_tune2.json 2>/dev/null
python bench.py --no-cpu-baseline > gpurun_out/sched/h_ship2.json 2>/dev/null
for f in c5_tune c5_load c5_tune2 h_tune h_ship h_tune2 h_ship2; do python - $f <<'PY'
import json,sys
d=json.loads(open(f'gpurun_out/sched/{sys.argv[1]}.json').read().strip().splitlines()[-1])
print(sys.argv[1], d['value'], d['ms_per_step'], d.get('single_stream_value'), d['roofline']['frac'], d['schedule'][:40])
PY
done
