#!/usr/bin/env python3
"""What a wider residual trunk would buy the bf16 path, priced on the CPU with the oracle's storage emulation
(oracle/forward.py: storage='bf16', trunk=None | 'f16' | 'f32'): mean / 99.9 % logit error of the three heads against the fp32
oracle, and how many of the fp32 oracle's detections the emulated heads reproduce after decode + NMS.

  python scripts/trunk_price.py [--size 608 --classes 80 --n 2]"""
import argparse, os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (os.path.join(ROOT, "yolo-v4-tf.keras_amd"), ROOT, os.path.join(ROOT, "tests")):
    sys.path.insert(0, p)
ap = argparse.ArgumentParser()
ap.add_argument("--size", type=int, default=608); ap.add_argument("--classes", type=int, default=80); ap.add_argument("--n", type=int, default=2)
ap.add_argument("--seed", type=int, default=0)
a = ap.parse_args()
from oracle import forward as OF
from oracle import decode_nms as ON
from yolo4hip import weights as W
from yolo4hip.plan import build_plan
from yolo4hip.config import make_config
ws = W.synth_weights(build_plan(a.size, a.classes), seed=a.seed)
imgs = W.synth_images(a.n, a.size, seed=a.seed)
cfg = make_config(a.size)
ref = OF.yolo_model_forward(imgs, ws, a.classes)
def dets(heads):
    return ON.decode_and_nms(heads, cfg, a.classes) if hasattr(ON, "decode_and_nms") else None
rows = []
for storage, trunk in (("bf16", None), ("bf16", "f16"), ("bf16", "f32"), ("f16", None)):
    emu = OF.yolo_model_forward(imgs, ws, a.classes, storage=storage, trunk=trunk)
    errs = [np.abs(e - r) for e, r in zip(emu, ref)]
    mean = [float(e.mean()) for e in errs]
    q = [float(np.quantile(e.ravel()[:: max(1, e.size // 2000000)], 0.999)) for e in errs]
    rows.append((storage, trunk, mean, q))
    print(f"storage {storage:5s} trunk {str(trunk):5s}: mean |err| per head {[round(m, 4) for m in mean]}  q99.9 {[round(x, 3) for x in q]}", flush=True)
