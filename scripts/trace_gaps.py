"""Inter-kernel gaps of one predict step from a rocprofv3 --kernel-trace CSV.
usage: trace_gaps.py <kernel_trace.csv> <out.json>
A step = the run of dispatches from one stem kernel (stem_down_kernel / stem_mfma_kernel / stem_conv_kernel) up to and
including the next nms_kernel.  Per step: sum of kernel durations, sum of gaps (start[i+1] - end[i], clamped at 0) between
consecutive dispatches, wall time first start -> last end.  Reported: the median step."""
import csv, json, statistics, sys

rows = []
for r in csv.DictReader(open(sys.argv[1])):
    rows.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"]))
rows.sort()
steps, cur = [], None
for s, e, name in rows:
    if "stem" in name and "pack" not in name:
        cur = []
    if cur is not None:
        cur.append((s, e, name))
        if "nms_kernel" in name:
            steps.append(cur)
            cur = None
out = []
for st in steps:
    busy = sum(e - s for s, e, _ in st)
    gaps = [max(0, st[i + 1][0] - st[i][1]) for i in range(len(st) - 1)]
    out.append({"launches": len(st), "kernel_us": busy / 1e3, "gap_us": sum(gaps) / 1e3, "wall_us": (st[-1][1] - st[0][0]) / 1e3,
                "max_gap_us": max(gaps) / 1e3, "median_gap_us": statistics.median(gaps) / 1e3})
out = out[1:] if len(out) > 2 else out           # drop the first profiled step (cold)
med = lambda k: statistics.median(o[k] for o in out)
res = {"steps": len(out), "launches_per_step": out[0]["launches"] if out else 0,
       "median_step": {k: round(med(k), 2) for k in ("kernel_us", "gap_us", "wall_us", "max_gap_us", "median_gap_us")} if out else {},
       "note": "gap = start[i+1] - end[i] between consecutive dispatches of a step, from rocprofv3 --kernel-trace timestamps (ns)"}
json.dump(res, open(sys.argv[2], "w"), indent=1)
print(res)
