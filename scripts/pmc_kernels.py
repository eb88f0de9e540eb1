"""Per-kernel means of rocprofv3 --pmc counters: usage pmc_kernels.py <out.json> <pass_dir> [<pass_dir> ...]
Each pass directory holds one rocprofv3 counter_collection CSV (one --pmc pass); counters are averaged per DISPATCH of a
kernel (template arguments kept, so each tile configuration is its own row)."""
import collections, csv, glob, json, os, sys

out = sys.argv[1]
agg = collections.defaultdict(lambda: collections.defaultdict(float))
cnt = collections.defaultdict(lambda: collections.defaultdict(int))
for d in sys.argv[2:]:
    for f in glob.glob(os.path.join(d, "**", "*counter_collection.csv"), recursive=True):
        for r in csv.DictReader(open(f)):
            k = r["Kernel_Name"].split("(")[0].strip()
            agg[k][r["Counter_Name"]] += float(r["Counter_Value"])
            cnt[k][r["Counter_Name"]] += 1
rows = {}
for k, v in agg.items():
    if "pack_" in k or "fold_bn" in k or "at::native" in k or "rocclr" in k:
        continue
    rows[k] = {c: val / cnt[k][c] for c, val in v.items()}
    rows[k]["dispatches"] = max(cnt[k].values())
json.dump(rows, open(out, "w"), indent=1, sort_keys=True)
for k, v in sorted(rows.items(), key=lambda kv: -kv[1].get("SQ_WAVE_CYCLES", 0) * kv[1]["dispatches"]):
    print(k[:100], {c: round(x, 1) for c, x in v.items()})
