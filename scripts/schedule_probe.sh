# the shipped schedule against a fresh autotune on this box, alternating (img/s, ms/step, single-stream img/s, conv-family frac)
mkdir -p gpurun_out/sched
for r in 1 2 3; do
python bench.py --retune --no-cpu-baseline --save-tiles gpurun_out/sched/retuned_$r.json > gpurun_out/sched/h_tune$r.json 2>/dev/null
python bench.py --no-cpu-baseline > gpurun_out/sched/h_ship$r.json 2>/dev/null
done
for f in h_tune1 h_ship1 h_tune2 h_ship2 h_tune3 h_ship3; do python - $f <<'PY'
import json,sys
d=json.loads(open(f'gpurun_out/sched/{sys.argv[1]}.json').read().strip().splitlines()[-1])
print(sys.argv[1], d['value'], d['ms_per_step'], d.get('single_stream_value'), d['roofline']['frac'], d['schedule'][:40])
PY
done
