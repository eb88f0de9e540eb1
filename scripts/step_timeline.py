"""Kernel timeline of ONE predict step from a rocprofv3 --kernel-trace CSV (the last complete step: from behind one nms_kernel up to and
including the next): per launch its start (us from the step's first kernel), duration and the gap to the previous launch, then the
durations summed per kernel.  usage: step_timeline.py <kernel_trace.csv>"""
import collections
import csv
import sys

rows = sorted((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"]) for r in csv.DictReader(open(sys.argv[1])))
idx = [i for i, r in enumerate(rows) if "nms_kernel" in r[2]]
st = rows[idx[-3] + 1:idx[-2] + 1]
t0 = st[0][0]
print("%d launches, %.1f us from the first kernel's start to the last one's end" % (len(st), (st[-1][1] - t0) / 1e3))
agg = collections.defaultdict(lambda: [0, 0.0])
for i, (s, e, n) in enumerate(st):
    k = n.split("(")[0]
    agg[k][0] += 1
    agg[k][1] += (e - s) / 1e3
    print("%3d %8.1f dur %6.1f gap %5.1f %s" % (i, (s - t0) / 1e3, (e - s) / 1e3, (s - st[i - 1][1]) / 1e3 if i else 0.0, k[:100]))
print()
for k, v in sorted(agg.items(), key=lambda kv: -kv[1][1]):
    print("%-100s %3d %8.1f" % (k[:100], v[0], v[1]))
