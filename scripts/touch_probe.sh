# A/B of the weight touch (conv_common.h: weight_touch; Y4_NO_WEIGHT_TOUCH=1 turns it off), shipped schedule, alternating
mkdir -p gpurun_out/touch
for r in 1 2 3; do for t in 1 0; do
  Y4_NO_WEIGHT_TOUCH=$t python bench.py --no-cpu-baseline --in-flight 1 > gpurun_out/touch/s1_$t.json 2>/dev/null
  Y4_NO_WEIGHT_TOUCH=$t python bench.py --no-cpu-baseline > gpurun_out/touch/s2_$t.json 2>/dev/null
  python - $t <<'PY'
import json,sys
d=json.loads(open(f'gpurun_out/touch/s1_{sys.argv[1]}.json').read().strip().splitlines()[-1])
e=json.loads(open(f'gpurun_out/touch/s2_{sys.argv[1]}.json').read().strip().splitlines()[-1])
print('no_touch', sys.argv[1], '| one stream', d['value'], d['ms_per_step'], d['roofline']['frac'], d['roofline']['kernel_ms_per_step'], '| two in flight', e['value'], e['ms_per_step'], e['single_stream_value'], e['outputs_sha256'][:12])
PY
done; done
