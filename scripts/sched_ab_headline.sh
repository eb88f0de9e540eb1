#!/bin/bash
# A/B of candidate headline schedules on ONE box, alternating: bench.py --load-tiles <file> for every file given, R rounds,
# two batches in flight (value) and one stream (single_stream_value, conv family ms).
R=${R:-3}
for r in $(seq 1 $R); do
  for f in "$@"; do
    python bench.py --no-cpu-baseline --no-latency --load-tiles $f --blocks 3 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read()); print('$f round $r: value', d['value'], 'one stream', d.get('single_stream_value'), 'conv ms', d['roofline']['kernel_ms_per_step'], 'backbone one-stream', d['roofline']['backbone_wall']['one_stream']['frac'])"
  done
done
