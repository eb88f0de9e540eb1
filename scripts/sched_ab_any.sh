#!/bin/bash
# usage: ab_any.sh "<bench args>" file1 file2 ...   (R rounds, alternating)
R=${R:-3}; ARGS=$1; shift
for r in $(seq 1 $R); do
  for f in "$@"; do
    python bench.py --no-cpu-baseline --no-latency --load-tiles $f --blocks 3 $ARGS 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read()); print('$f round $r: value', d['value'], 'one stream', d.get('single_stream_value'), 'conv ms', d['roofline']['kernel_ms_per_step'])"
  done
done
