"""The oracle itself (oracle/decode_nms.py, oracle/forward.py) against hand-computable known answers and
the committed golden fixtures.  The reference has no tests or vectors for this path (parity unpinned), so
the known answers below are derived by hand from the cited reference lines."""
import os

import numpy as np
import pytest

from helpers import GOLDEN
from oracle import decode_nms as OD

ANCHORS = [12, 16, 19, 36, 40, 28, 36, 75, 76, 55, 72, 146, 142, 110, 192, 243, 459, 401]
XYSCALE = [1.2, 1.1, 1.05]


def test_decode_zero_logits_closed_form():
    """custom_layers.py:251-256 with t = 0: sigmoid = 0.5 -> xy = (0.5*s - 0.5*(s-1) + grid)*stride
    = (grid + 0.5)*stride; wh = anchor; obj = cls = 0.5."""
    for scale, (g, stride) in enumerate(((13, 8), (7, 16), (4, 32))):
        pred = np.zeros((2, g, g, 3 * 7), np.float32)
        anc = np.asarray(ANCHORS, np.float32).reshape(3, 3, 2)[scale]
        box, obj, cls, xywh = OD.get_boxes(pred, anc, 2, g, stride, XYSCALE[scale])
        assert box.shape == (2, g, g, 3, 4) and obj.shape == (2, g, g, 3, 1) and cls.shape == (2, g, g, 3, 2)
        assert np.all(obj == 0.5) and np.all(cls == 0.5)
        for row in (0, g - 1):
            for col in (0, g // 2):
                for a in range(3):
                    cx, cy = (col + 0.5) * stride, (row + 0.5) * stride
                    want = [cx - anc[a, 0] / 2, cy - anc[a, 1] / 2, cx + anc[a, 0] / 2, cy + anc[a, 1] / 2]
                    assert np.allclose(box[1, row, col, a], want, atol=1e-4), (scale, row, col, a)


def test_decode_single_hot_logit_and_channel_order():
    """channel = anchor*(5+C) + field, fields x,y,w,h,obj,cls...; grid[i,j] = (x=j, y=i) (:247-249)."""
    g, stride, C = 5, 16, 3
    pred = np.zeros((1, g, g, 3 * (5 + C)), np.float32)
    row, col, a = 3, 1, 2
    base = a * (5 + C)
    pred[0, row, col, base + 0] = 2.0       # tx
    pred[0, row, col, base + 3] = np.log(2.0)   # th -> doubles the anchor height
    pred[0, row, col, base + 4] = 5.0       # obj
    pred[0, row, col, base + 5 + 1] = -5.0  # class 1
    anc = np.asarray(ANCHORS, np.float32).reshape(3, 3, 2)[1]
    box, obj, cls, _ = OD.get_boxes(pred, anc, C, g, stride, 1.1)
    sx = 1 / (1 + np.exp(-2.0))
    cx = (sx * 1.1 - 0.05 + col) * stride
    cy = (0.5 * 1.1 - 0.05 + row) * stride
    w, h = anc[a, 0], anc[a, 1] * 2
    assert np.allclose(box[0, row, col, a], [cx - w / 2, cy - h / 2, cx + w / 2, cy + h / 2], atol=1e-3)
    assert abs(obj[0, row, col, a, 0] - 1 / (1 + np.exp(-5.0))) < 1e-6
    assert abs(cls[0, row, col, a, 1] - 1 / (1 + np.exp(5.0))) < 1e-6
    assert np.all(obj[0, 0, 0] == 0.5)


def test_flatten_order_and_box_index_convention():
    """nms() flattens each scale in (row, col, anchor) order and concatenates scales 0,1,2
    (custom_layers.py:273-280): box index n = off_s + (row*g + col)*3 + a; boxes / input size (:284)."""
    size, C = 64, 2
    outs = [np.zeros((1, size // s, size // s, 3 * (5 + C)), np.float32) for s in (8, 16, 32)]
    outs[1][0, 2, 1, 1 * (5 + C) + 4] = 9.0       # scale 1 (g=4), row 2, col 1, anchor 1: high objectness
    outs[1][0, 2, 1, 1 * (5 + C) + 5] = 9.0       # class 0
    head = OD.yolov4_head(outs, C, ANCHORS, XYSCALE)
    boxes, scores = OD.flatten_for_nms(head, size, C)
    nb = 3 * (64 + 16 + 4)
    assert boxes.shape == (1, nb, 4) and scores.shape == (1, nb, C)
    n = 3 * 64 + (2 * 4 + 1) * 3 + 1
    assert np.argmax(scores[0, :, 0]) == n and scores[0, n, 0] > 0.99
    cx, cy = (1 + 0.5) * 16 / size, (2 + 0.5) * 16 / size
    assert np.allclose(boxes[0, n], [cx - 76 / 2 / size, cy - 55 / 2 / size, cx + 76 / 2 / size, cy + 55 / 2 / size], atol=1e-5)


def _nms(boxes, scores, **kw):
    return OD.combined_nms(np.asarray(boxes, np.float32)[None], np.asarray(scores, np.float32)[None], **kw)


def test_nms_strict_thresholds():
    """score == threshold is rejected (strict '>'); IoU == threshold is KEPT (suppress only when '>')."""
    boxes = [[0, 0, 0.5, 0.5], [0.6, 0.6, 0.9, 0.9]]
    b, s, c, v, i = _nms(boxes, [[np.float32(0.3)], [0.31]])
    assert v[0] == 1 and i[0, 0] == 1
    # two boxes with IoU exactly 1/3: [0,0,1,0.5] vs [0,0.25,1,0.75] -> inter 0.25, union 0.75
    boxes = [[0, 0, 0.5, 1.0], [0.25, 0, 0.75, 1.0]]
    thr = float(np.float32(0.25) / np.float32(0.75))
    b, s, c, v, i = _nms(boxes, [[0.9], [0.8]], iou_threshold=thr)
    assert v[0] == 2                                  # IoU == thr -> not suppressed
    b, s, c, v, i = _nms(boxes, [[0.9], [0.8]], iou_threshold=thr - 1e-3)
    assert v[0] == 1 and i[0, 0] == 0


def test_nms_is_class_aware_and_sorted():
    boxes = [[0.1, 0.1, 0.5, 0.5], [0.1, 0.1, 0.5, 0.5], [0.6, 0.6, 0.9, 0.9]]
    scores = [[0.9, 0.0], [0.0, 0.8], [0.5, 0.95]]
    b, s, c, v, i = _nms(boxes, scores)
    assert v[0] == 4                                   # identical boxes of DIFFERENT classes are both kept
    assert list(s[0, :4]) == sorted(s[0, :4], reverse=True)
    assert list(i[0, :4]) == [2, 0, 1, 2] and list(c[0, :4]) == [1, 0, 1, 0]
    assert np.all(b[0, 4:] == 0) and np.all(s[0, 4:] == 0) and np.all(i[0, 4:] == -1)


def test_nms_tie_order_is_index_then_class():
    boxes = [[0.0, 0.0, 0.2, 0.2], [0.4, 0.4, 0.6, 0.6], [0.7, 0.7, 0.9, 0.9]]
    scores = [[0.5, 0.5], [0.5, 0.5], [0.5, 0.5]]
    b, s, c, v, i = _nms(boxes, scores)
    assert list(i[0, :6]) == [0, 0, 1, 1, 2, 2] and list(c[0, :6]) == [0, 1, 0, 1, 0, 1]


def test_golden_nms_cases_and_their_meaning():
    g = np.load(os.path.join(GOLDEN, "nms_cases.npz"))
    for tag in "ABC":
        r = OD.combined_nms(g[f"{tag}_boxes"], g[f"{tag}_scores"])
        for name, arr in zip(("ob", "os", "oc", "ov", "oi"), r):
            assert np.array_equal(arr, g[f"{tag}_{name}"]), (tag, name)
    assert g["B_ov"][0] == 100 and np.all(g["B_oc"][0] == 1)          # 150 separated boxes -> capped at 100
    assert list(g["B_oi"][0][:3]) == [0, 1, 2]
    # C: clip to [0,1]; zero-area box never suppresses nor is suppressed (IoU 0); reversed corners are normalised
    assert g["C_ov"][0] == 4
    assert np.allclose(g["C_ob"][0, 0], [0, 0, 0.3, 0.4])
    assert list(g["C_oi"][0][:4]) == [0, 1, 2, 4]                        # box 3 is suppressed by box 2 (same region)
    assert np.allclose(g["C_ob"][0, 3], [0.7, 0.7, 1.0, 1.0])


def test_golden_tiny_net_regression():
    """The oracle forward + decode + NMS reproduces the committed fixture (guards against silent oracle
    drift: the GPU tests compare the HIP path with the same fixture)."""
    from golden.make_golden import TINY, sample_idx  # noqa: F401
    from yolo4hip import weights as W
    from yolo4hip.config import make_config
    from yolo4hip.plan import build_plan
    from oracle import forward as OF
    g = np.load(os.path.join(GOLDEN, "tiny_net.npz"))
    for size, ncls, n, seed in TINY:
        cfg = make_config(size)
        plan = build_plan(size, ncls)
        heads = OF.yolo_model_forward(W.synth_images(n, size, seed), W.synth_weights(plan, seed), ncls)
        tag = f"s{size}c{ncls}"
        for i, h in enumerate(heads):
            assert np.allclose(h.reshape(-1)[g[f"{tag}_head{i}_idx"]], g[f"{tag}_head{i}_val"], atol=2e-4)
            assert abs(np.abs(h.astype(np.float64)).sum() - g[f"{tag}_head{i}_abssum"]) < 1e-3 * g[f"{tag}_head{i}_abssum"]
        b, s, c, v, k = OD.inference_from_heads(heads, ncls, cfg["anchors"], cfg["xyscale"], size, score_threshold=0.05)
        assert np.array_equal(v, g[f"{tag}_valid"])
        assert (k == g[f"{tag}_kept"]).mean() > 0.97        # near-ties may swap across BLAS builds
        assert np.abs(s - g[f"{tag}_scores"]).max() < 1e-3
        assert v.min() > 0


def test_conv_block_matches_direct_definition():
    """oracle conv(): stride-2 = top/left zero pad + 'valid' (custom_layers.py:9-12); BN eps 1e-3."""
    from oracle.forward import conv_block
    from yolo4hip.weights import ConvWeights
    rng = np.random.default_rng(0)
    x = rng.standard_normal((1, 6, 6, 2)).astype(np.float32)
    w = rng.standard_normal((3, 2, 3, 3)).astype(np.float32)
    bn = np.stack([np.full(3, 0.1), np.full(3, 2.0), np.full(3, 0.5), np.full(3, 4.0)]).astype(np.float32)
    y = conv_block(x, ConvWeights(w=w, bn=bn), 3, 2, None)
    assert y.shape == (1, 3, 3, 3)
    xp = np.zeros((8, 8, 2), np.float32); xp[1:7, 1:7] = x[0]          # pad top/left 1 (bottom/right unused)
    for (oy, ox, co) in ((0, 0, 0), (2, 1, 2), (1, 2, 1)):
        patch = xp[2 * oy:2 * oy + 3, 2 * ox:2 * ox + 3]                # rows 2oy-1..2oy+1 of the unpadded input
        acc = sum(patch[ky, kx, ci] * w[co, ci, ky, kx] for ky in range(3) for kx in range(3) for ci in range(2))
        want = (acc - 0.5) * 2.0 / np.sqrt(4.0 + 1e-3) + 0.1
        assert abs(y[0, oy, ox, co] - want) < 1e-4
