"""Summarise rocprofv3 counter_collection CSVs (one directory per --pmc pass) into per-kernel totals per step.

usage: pmc_summary.py <steps_profiled> <out.json> <pass_dir> [<pass_dir> ...]
FETCH_SIZE / WRITE_SIZE are in the tool's KB; MI355X_MICROARCH.md (HBM section): on gfx950 FETCH_SIZE counts 64 B per
128 B read request, so corrected HBM read bytes = 2 * FETCH_SIZE (applied in the 'hbm_bytes_per_step' fields).
"""
import collections, csv, glob, json, os, re, sys

steps, out = int(sys.argv[1]), sys.argv[2]
agg = collections.defaultdict(lambda: collections.defaultdict(float))
launches = collections.Counter()
for d in sys.argv[3:]:
    for f in glob.glob(os.path.join(d, "**", "*counter_collection.csv"), recursive=True):
        seen = set()
        for r in csv.DictReader(open(f)):
            k = re.sub(r"\(.*", "", r["Kernel_Name"]).strip()
            agg[k][r["Counter_Name"]] += float(r["Counter_Value"])
            if (r["Dispatch_Id"], d) not in seen and r["Counter_Name"] in ("FETCH_SIZE", "SQ_WAVES"):
                seen.add((r["Dispatch_Id"], d))
                launches[k] += 1
rows = []
tot_f = tot_w = 0.0
for k, v in sorted(agg.items(), key=lambda kv: -(kv[1].get("FETCH_SIZE", 0) + kv[1].get("WRITE_SIZE", 0))):
    if "pack_" in k or "fold_bn" in k:
        continue                      # one-off weight packing, not part of a step
    row = {"kernel": k, "launches_per_step": launches[k] / steps}
    for c, val in v.items():
        row[c + "_per_step"] = val / steps
    if "FETCH_SIZE" in v or "WRITE_SIZE" in v:
        row["hbm_read_bytes_per_step"] = 2 * v.get("FETCH_SIZE", 0.0) * 1024 / steps
        row["hbm_write_bytes_per_step"] = v.get("WRITE_SIZE", 0.0) * 1024 / steps
        if any(x in k for x in ("conv_igemm", "conv_halo", "csp_stage", "resblock", "conv_p8", "conv_l12")):      # the conv kernel family
            tot_f += row["hbm_read_bytes_per_step"]; tot_w += row["hbm_write_bytes_per_step"]
    rows.append(row)
json.dump({"config": {"size": 608, "classes": 80, "batch": 32, "dtype": "bf16", "stem_fusion": True, "chain_fusion": True,
                      "stage_fusion": any("csp_stage" in r["kernel"] for r in rows)},
           "steps_profiled": steps, "conv_igemm_hbm_read_bytes_per_step": tot_f, "conv_igemm_hbm_write_bytes_per_step": tot_w,
           "conv_igemm_hbm_bytes_per_step": tot_f + tot_w, "kernels": rows}, open(out, "w"), indent=1)
print("conv_igemm family: read %.3f GB + write %.3f GB per step" % (tot_f / 1e9, tot_w / 1e9))
