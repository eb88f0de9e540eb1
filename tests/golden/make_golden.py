#!/usr/bin/env python3
"""Regenerates the committed fixtures in tests/golden/ from the CPU oracle (oracle/).

No reference-generated vectors exist (the reference has no tests and TensorFlow cannot be imported in the
build container -- SURVEY.md §8c), so these are ORACLE-generated regression pins, plus the reference's own
demo input `street.jpeg` (reference img/street.jpeg, 273x185) used by the predict() plumbing tests.

  python tests/golden/make_golden.py
writes
  tiny_net.npz   seeded synthetic weights/images (regenerable from the seed, not stored) -> per-head
                 checksums, 256 sampled logits per head, and the 4 NMS outputs + kept indices, for
                 (size 96, 2 classes, batch 2) and (size 160, 3 classes, batch 1)
  nms_cases.npz  hand-built NMS inputs (boxes, scores) with the oracle's outputs
"""
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, os.path.join(ROOT, "yolo-v4-tf.keras_amd"))
sys.path.insert(0, ROOT)

from yolo4hip import weights as W  # noqa: E402
from yolo4hip.config import make_config  # noqa: E402
from yolo4hip.plan import build_plan  # noqa: E402
from oracle import forward as OF, decode_nms as OD  # noqa: E402

TINY = [(96, 2, 2, 7), (160, 3, 1, 8)]     # size, classes, batch, seed


def sample_idx(n, k=256):
    return np.random.default_rng(123).choice(n, size=min(k, n), replace=False)


def tiny_net():
    out = {}
    for size, ncls, n, seed in TINY:
        cfg = make_config(size)
        plan = build_plan(size, ncls)
        ws = W.synth_weights(plan, seed)
        imgs = W.synth_images(n, size, seed)
        heads = OF.yolo_model_forward(imgs, ws, ncls)
        tag = f"s{size}c{ncls}"
        for i, h in enumerate(heads):
            flat = h.reshape(-1)
            idx = sample_idx(flat.size)
            out[f"{tag}_head{i}_sum"] = np.float64(flat.astype(np.float64).sum())
            out[f"{tag}_head{i}_abssum"] = np.float64(np.abs(flat.astype(np.float64)).sum())
            out[f"{tag}_head{i}_idx"] = idx
            out[f"{tag}_head{i}_val"] = flat[idx]
        # a lower score threshold so the tiny nets produce detections
        b, s, c, v, k = OD.inference_from_heads(heads, ncls, cfg["anchors"], cfg["xyscale"], size,
                                                score_threshold=0.05)
        out[f"{tag}_boxes"], out[f"{tag}_scores"], out[f"{tag}_classes"] = b, s, c
        out[f"{tag}_valid"], out[f"{tag}_kept"] = v, k
    np.savez_compressed(os.path.join(HERE, "tiny_net.npz"), **out)


def nms_cases():
    rng = np.random.default_rng(42)
    cases = {}
    # A: 40 boxes, 3 classes, clustered so that suppression happens
    centers = rng.random((8, 2)) * 0.8 + 0.1
    boxes = []
    for c in centers:
        for _ in range(5):
            ctr = c + rng.normal(0, 0.01, 2)
            wh = 0.1 + rng.random(2) * 0.05
            boxes.append([ctr[0] - wh[0] / 2, ctr[1] - wh[1] / 2, ctr[0] + wh[0] / 2, ctr[1] + wh[1] / 2])
    cases["A_boxes"] = np.asarray(boxes, np.float32)[None]
    cases["A_scores"] = rng.random((1, 40, 3)).astype(np.float32)
    # B: 150 well-separated boxes in one class, all above threshold -> the 100 cap binds
    g = np.stack(np.meshgrid(np.arange(15), np.arange(10)), -1).reshape(-1, 2).astype(np.float32)
    bb = np.concatenate([g * 0.06 + 0.01, g * 0.06 + 0.05], -1)
    cases["B_boxes"] = bb[None].astype(np.float32)
    sc = np.zeros((1, 150, 2), np.float32)
    sc[0, :, 1] = np.linspace(0.95, 0.35, 150, dtype=np.float32)
    cases["B_scores"] = sc
    # C: boxes sticking out of [0,1] (clip), a degenerate zero-area box, reversed corners
    cases["C_boxes"] = np.asarray([[[-0.2, -0.1, 0.3, 0.4], [0.5, 0.5, 0.5, 0.9], [0.9, 0.9, 0.6, 0.6],
                                    [0.58, 0.58, 0.93, 0.93], [0.7, 0.7, 1.4, 1.2]]], np.float32)
    cases["C_scores"] = np.asarray([[[0.9, 0.0], [0.8, 0.0], [0.7, 0.0], [0.6, 0.0], [0.0, 0.5]]], np.float32)
    for tag in "ABC":
        r = OD.combined_nms(cases[f"{tag}_boxes"], cases[f"{tag}_scores"])
        for name, arr in zip(("ob", "os", "oc", "ov", "oi"), r):
            cases[f"{tag}_{name}"] = arr
    np.savez_compressed(os.path.join(HERE, "nms_cases.npz"), **cases)


if __name__ == "__main__":
    tiny_net()
    nms_cases()
    print("wrote", sorted(f for f in os.listdir(HERE) if f.endswith(".npz")))
