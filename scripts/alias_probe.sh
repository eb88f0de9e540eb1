set -x
mkdir -p gpurun_out/alias
python -m pytest tests/test_gpu_api.py -x -q -k "aliasing or in_flight or autotune_pair" 2>&1 | tail -5
python bench.py --no-alias-workspace --save-tiles gpurun_out/alias/tiles.json > gpurun_out/alias/plain2.json 2> gpurun_out/alias/plain2.err
python bench.py --load-tiles gpurun_out/alias/tiles.json --alias-workspace > gpurun_out/alias/alias2.json 2> gpurun_out/alias/alias2.err
python bench.py --load-tiles gpurun_out/alias/tiles.json --in-flight 1 --no-alias-workspace > gpurun_out/alias/plain1.json 2>/dev/null
python bench.py --load-tiles gpurun_out/alias/tiles.json --in-flight 1 --alias-workspace > gpurun_out/alias/alias1.json 2>/dev/null
python bench.py --load-tiles gpurun_out/alias/tiles.json --alias-workspace --in-flight 3 > gpurun_out/alias/alias3.json 2>/dev/null
python bench.py --no-alias-workspace --load-tiles gpurun_out/alias/tiles.json > gpurun_out/alias/plain2b.json 2>/dev/null
python bench.py --load-tiles gpurun_out/alias/tiles.json --alias-workspace > gpurun_out/alias/alias2b.json 2>/dev/null
python bench.py --alias-workspace > gpurun_out/alias/alias2_owntune.json 2>/dev/null
for f in plain2 alias2 plain1 alias1 alias3 plain2b alias2b alias2_owntune; do python - $f <<'PY'
import json,sys
d=json.loads(open(f'gpurun_out/alias/{sys.argv[1]}.json').read().strip().splitlines()[-1])
print(sys.argv[1], d['value'], d['ms_per_step'], d.get('single_stream_value'), d['roofline']['frac'], d.get('activation_workspace_bytes'), d['outputs_sha256'][:12])
PY
done
