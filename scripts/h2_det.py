#!/usr/bin/env python3
"""Determinism hunt for the halo2 tiles inside the engine: 608/80/n4 with the shipped schedule (optionally with the halo2 ids replaced),
forwards under a side stream's load; prints the first differing tap of every forward that differs.
  python scripts/h2_det.py [--dtype f16] [--forwards 60] [--swap 61:55,62:57]"""
import argparse, json, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (os.path.join(ROOT, "yolo-v4-tf.keras_amd"), ROOT, os.path.join(ROOT, "tests")):
    sys.path.insert(0, p)
import torch
ap = argparse.ArgumentParser()
ap.add_argument("--dtype", default="f16"); ap.add_argument("--forwards", type=int, default=60); ap.add_argument("--swap", default="")
ap.add_argument("--n", type=int, default=4)
a = ap.parse_args()
from test_gpu_determinism import _engine, _inputs, tap_snapshot, first_tap_difference
size, ncls, n = 608, 80, a.n
eng = _engine(size, ncls, n, a.dtype, seed=1)
eng.set_stem_fusion(True); eng.set_chain_fusion(True); eng.set_stage_fusion(True); eng.set_res_fusion(True)
sched = os.path.join(ROOT, "yolo-v4-tf.keras_amd", "yolo4hip", "schedules", f"608_80_32_{a.dtype}.json")
saved = json.load(open(sched))
swap = dict((int(x.split(":")[0]), int(x.split(":")[1])) for x in a.swap.split(",") if x)
saved["tiles"] = [swap.get(t, t) for t in saved["tiles"]]
eng.apply_schedule(saved)
print("halo2 ids in use:", sorted(set(t for t in saved["tiles"] if 55 <= t <= 62)))
fl, u8 = _inputs(eng, size, n)
side = torch.cuda.Stream(device=eng.device)
junk = torch.randn(2048, 2048, device=eng.device)
base, bad = None, 0
for r in range(a.forwards):
    if r % 2:
        with torch.cuda.stream(side):
            for _ in range(4):
                junk = (junk @ junk).clamp_(-1, 1)
    eng.forward_device(u8 if r % 2 else fl)
    snap = tap_snapshot(eng, n)
    if base is None:
        base = snap
        continue
    diff = first_tap_difference(base, snap, eng)
    if diff is not None:
        bad += 1
        print(f"forward {r}: {diff}", flush=True)
side.synchronize()
print(f"{bad} of {a.forwards - 1} forwards differ")
