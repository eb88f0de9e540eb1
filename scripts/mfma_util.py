"""MFMA utilisation and SQ wave-cycle breakdown per kernel from the --pmc passes summarised by pmc_kernels.py.
usage: mfma_util.py <pmc_summary.json> <kernel_stats.csv or -> <out_mfma_util.json> <out_sq_wait.json>
  mfma_util     = SQ_VALU_MFMA_BUSY_CYCLES / (GRBM_GUI_ACTIVE / 8 XCDs * 256 CUs * 4 SIMDs)
                  (the gfx94x derived-metric formula MfmaUtil; SQ_VALU_MFMA_BUSY_CYCLES counts pipe cycles summed over
                  SIMDs: 16 per v_mfma_f32_16x16x32, GRBM_GUI_ACTIVE is summed over the 8 XCDs by rocprofv3)
  parked / stalled / active = SQ_WAIT_ANY, SQ_WAIT_INST_ANY, SQ_ACTIVE_INST_ANY over SQ_WAVE_CYCLES (disjoint, MI355X_MICROARCH.md)
"""
import json, sys
pm = json.load(open(sys.argv[1]))
util, wait = {}, {}
for k, v in pm.items():
    if "SQ_VALU_MFMA_BUSY_CYCLES" in v and "GRBM_GUI_ACTIVE" in v and v["GRBM_GUI_ACTIVE"] > 0:
        util[k] = {"mfma_util": round(v["SQ_VALU_MFMA_BUSY_CYCLES"] / (v["GRBM_GUI_ACTIVE"] / 8 * 1024), 4),
                   "SQ_VALU_MFMA_BUSY_CYCLES": v["SQ_VALU_MFMA_BUSY_CYCLES"], "GRBM_GUI_ACTIVE": v["GRBM_GUI_ACTIVE"],
                   "SQ_BUSY_CYCLES": v.get("SQ_BUSY_CYCLES"), "dispatches_profiled": v["dispatches"]}
        if "SQ_INSTS_VALU_MFMA_MOPS_BF16" in v:
            util[k]["SQ_INSTS_VALU_MFMA_MOPS_BF16"] = v["SQ_INSTS_VALU_MFMA_MOPS_BF16"]
    if "SQ_WAVE_CYCLES" in v and v["SQ_WAVE_CYCLES"] > 0:
        wc = v["SQ_WAVE_CYCLES"]
        wait[k] = {"parked_frac": round(v.get("SQ_WAIT_ANY", 0) / wc, 4), "issue_stall_frac": round(v.get("SQ_WAIT_INST_ANY", 0) / wc, 4),
                   "active_frac": round(v.get("SQ_ACTIVE_INST_ANY", 0) / wc, 4),
                   "valu_active_frac": round(v.get("SQ_ACTIVE_INST_VALU", 0) / wc, 4) if "SQ_ACTIVE_INST_VALU" in v else None,
                   "lds_bank_conflict_cycles": v.get("SQ_LDS_BANK_CONFLICT"), "SQ_WAVE_CYCLES": wc, "waves": v.get("SQ_WAVES")}
json.dump({"formula": "SQ_VALU_MFMA_BUSY_CYCLES / (GRBM_GUI_ACTIVE / 8 * 256 * 4)", "kernels": util}, open(sys.argv[3], "w"), indent=1, sort_keys=True)
json.dump({"note": "fractions of SQ_WAVE_CYCLES; parked = s_waitcnt/barrier, issue_stall = dependency/pipe", "kernels": wait}, open(sys.argv[4], "w"), indent=1, sort_keys=True)
for k, v in sorted(util.items(), key=lambda kv: -kv[1]["SQ_VALU_MFMA_BUSY_CYCLES"] * kv[1]["dispatches_profiled"])[:14]:
    print("%-70s mfma_util %.3f  parked %.2f stall %.2f active %.2f" % (k[-70:], v["mfma_util"], wait.get(k, {}).get("parked_frac", 0),
          wait.get(k, {}).get("issue_stall_frac", 0), wait.get(k, {}).get("active_frac", 0)))
