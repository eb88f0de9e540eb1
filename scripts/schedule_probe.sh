# schedules against each other on one box, alternating: the shipped one, a candidate file ($1, e.g. scripts/vote_schedule.py's output copied
# somewhere gpurun pushes, such as scratch/) and a fresh autotune (img/s two in flight, ms/step, single-stream img/s, conv-family frac)
CAND=${1:-}
mkdir -p gpurun_out/sched
for r in 1 2 3; do
python bench.py --no-cpu-baseline > gpurun_out/sched/h_ship$r.json 2>/dev/null
if [ -n "$CAND" ]; then python bench.py --no-cpu-baseline --load-tiles $CAND > gpurun_out/sched/h_cand$r.json 2>/dev/null; fi
python bench.py --retune --no-cpu-baseline --save-tiles gpurun_out/sched/retuned_$r.json > gpurun_out/sched/h_tune$r.json 2>/dev/null
done
for f in h_ship1 h_cand1 h_tune1 h_ship2 h_cand2 h_tune2 h_ship3 h_cand3 h_tune3; do [ -f gpurun_out/sched/$f.json ] && python - $f <<'PY'
import json,sys
d=json.loads(open(f'gpurun_out/sched/{sys.argv[1]}.json').read().strip().splitlines()[-1])
print(sys.argv[1], d['value'], d['ms_per_step'], d.get('single_stream_value'), d['roofline']['frac'], d['schedule'][:40])
PY
done
