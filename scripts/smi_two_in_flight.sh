# rocm-smi (sclk, power, temperature) in the middle of a steady load with TWO batches in flight, then with one, on the same box
mkdir -p gpurun_out/smi2
for d in 2 1; do
python bench.py --no-cpu-baseline --in-flight $d --steps 1500 --blocks 5 > gpurun_out/smi2/bench_if$d.json 2>/dev/null &
BPID=$!
sleep 20
( for i in $(seq 1 16); do rocm-smi --showclocks --showpower --showtemp 2>/dev/null | grep -E "sclk|Power \(W\)|junction"; sleep 0.4; done ) > gpurun_out/smi2/smi_if$d.txt 2>&1
wait $BPID
python - $d <<'PY'
import json,sys,re
d=json.loads(open(f'gpurun_out/smi2/bench_if{sys.argv[1]}.json').read().strip().splitlines()[-1])
t=open(f'gpurun_out/smi2/smi_if{sys.argv[1]}.txt').read()
clk=[int(x) for x in re.findall(r'sclk clock level: \d+: \((\d+)Mhz\)',t)]; pw=[float(x) for x in re.findall(r'Power \(W\): ([\d.]+)',t)]; tj=[float(x) for x in re.findall(r'junction\) \(C\): ([\d.]+)',t)]
print('in flight', sys.argv[1], d['value'], 'img/s', d['ms_per_step'], 'ms | sclk', min(clk), max(clk), '| W', min(pw), max(pw), '| Tj', min(tj), max(tj))
PY
done
rocm-smi --showmaxpower 2>/dev/null | grep -i "max"
