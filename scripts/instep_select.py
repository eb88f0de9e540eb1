"""Does choosing tiles by their IN-STEP time (inputs and weights as a step leaves them) beat the tuner's hot-loop choice?
Several hot-tuned schedules for the headline shape (the shipped one + fresh `autotune` runs on this box) are each profiled
per op inside whole steps; a combined schedule takes, launch by launch, the variant with the lowest in-step time, and all
of them are then timed as whole single-stream steps, alternating.  usage: instep_select.py [n_retunes=3]
Writes the combination as gpurun_out/instep_combo.json in the format of yolo4hip/schedules/*.json.
Measured (round 3, two boxes): the combination is no better than the individual hot-tuned schedules (5.534-5.548 against
5.536-5.548 ms on one box; 5.684-5.691 against 5.646-5.684 on another) -- per-op event times inside a step are too noisy to
beat the tuner's head-to-head blocks, and since the weight touch a launch in a step runs within ~10 % of its hot-loop time.
The shipped schedule therefore stays a plain tuning run."""
import sys, os, json
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "yolo-v4-tf.keras_amd")); sys.path.insert(0, ROOT)
import numpy as np, torch
from yolo4hip import weights as W
from yolo4hip.config import make_config
from yolo4hip.engine import Engine
from yolo4hip.plan import build_plan

NR = int(sys.argv[1]) if len(sys.argv) > 1 else 3
size, ncls, n = 608, 80, 32
eng = Engine(ncls, make_config(size), max_batch=n, dtype="bf16", alias_workspace=True)
eng.load_weight_blob(W.flatten(W.synth_weights(build_plan(size, ncls), 0)))
imgs = torch.from_numpy(W.synth_images(n, size, seed=0)).to(eng.device)
outs = eng.alloc_outputs(n)


def fusions_on():
    eng.set_stem_fusion(True); eng.set_chain_fusion(True); eng.set_stage_fusion(True); eng.set_res_fusion(True)


fusions_on()
eng.predict_device(imgs, outs)
scheds = [dict(eng.shipped_schedule(), name="shipped")]
for r in range(NR):
    fusions_on()
    tiles = eng.autotune(n, reps=3)
    # (the fusion-kernel switches are taken from the shipped schedule, so that the op rows of all schedules mean the same)
    scheds.append({"tiles": tiles, "stage_fusion": scheds[0]["stage_fusion"], "res_fusion_mask": scheds[0]["res_fusion_mask"], "name": f"tune{r}"})


def apply(s):
    fusions_on()
    eng.apply_schedule(s)


def perop(s, reps=8):
    apply(s)
    eng.predict_device(imgs, outs)
    acc = None
    for _ in range(reps):
        rows = eng.profile(imgs)
        acc = [(nm, ms) for nm, ms in rows] if acc is None else [(nm, a + ms) for (nm, a), (_, ms) in zip(acc, rows)]
    return [(nm, a / reps) for nm, a in acc]


def step_ms(s, steps=60):
    apply(s)
    for _ in range(5): eng.predict_device(imgs, outs)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(steps): eng.predict_device(imgs, outs)
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / steps


# only schedules with the same fusion switches can be mixed launch by launch (the op rows then mean the same)
base = scheds[0]
same = [s for s in scheds if s["stage_fusion"] == base["stage_fusion"] and s["res_fusion_mask"] == base["res_fusion_mask"]]
tables = [perop(s) for s in same]
names = [nm for nm, _ in tables[0]]
assert all([nm for nm, _ in t] == names for t in tables), "op rows differ between schedules"
# a row 'cA' or 'cA+B' covers conv A (and what is fused behind it): take conv A's tile entry, and the entries of every conv up
# to the next row's first conv, from the schedule whose row is fastest
firsts = []
for nm in names:
    firsts.append(int(nm[1:].split("+")[0]) if nm.startswith("c") and nm[1:2].isdigit() else None)
combo = list(base["tiles"])
picked = {}
for i, nm in enumerate(names):
    if firsts[i] is None: continue
    nxt = next((f for f in firsts[i + 1:] if f is not None), 110)
    best = min(range(len(same)), key=lambda k: tables[k][i][1])
    picked[nm] = (same[best]["name"], [round(t[i][1] * 1e3, 1) for t in tables])
    for c in range(firsts[i], nxt):
        combo[c] = same[best]["tiles"][c]
cs = {"tiles": combo, "stage_fusion": base["stage_fusion"], "res_fusion_mask": base["res_fusion_mask"], "name": "in-step combination"}
gain = sum(min(t[i][1] for t in tables) for i in range(len(names))), [sum(ms for _, ms in t) for t in tables]
print("sum of per-op rows (ms): best-of", round(gain[0], 4), "schedules", [round(g, 4) for g in gain[1]])
for rnd in range(3):
    print("round", rnd, " | ".join(f"{s['name']} {step_ms(s):.4f} ms" for s in same + [cs]), flush=True)
differs = [(nm, v) for nm, v in picked.items() if max(v[1]) - min(v[1]) > 1.5]
print("rows where the schedules differ by > 1.5 us in step:", differs[:40])
out = {"size": size, "classes": ncls, "batch": n, "dtype": "bf16", "tiles": [int(t) for t in combo], "stage_fusion": bool(base["stage_fusion"]),
       "res_fusion_mask": int(base["res_fusion_mask"]), "in_flight": int(base.get("in_flight", 2))}
json.dump(out, open(os.path.join(ROOT, "gpurun_out", "instep_combo.json"), "w"))
