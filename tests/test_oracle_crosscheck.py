"""Independent cross-checks of the oracle (parity is unpinned against TensorFlow -- LABNOTES.md section 2 -- so the
restatement is at least checked against SECOND implementations of the same published semantics):
  * `oracle.forward.conv_block` against torch.nn modules in eval mode (Conv2d / BatchNorm2d(eps=1e-3) / Mish /
    LeakyReLU(0.1) / ZeroPad2d((1,0,1,0))): the library's own statement of Keras' Conv2D -> BatchNormalization ->
    activation (reference custom_layers.py:5-31), incl. the top/left-only padding of the stride-2 convs;
  * `oracle.decode_nms.combined_nms` against a brute-force per-class greedy NMS written from the textbook definition
    (sort by score, keep, discard IoU > threshold, per class; then the global top-k), on random boxes via hypothesis.
"""
import numpy as np
import pytest
import torch

from oracle import decode_nms as OD, forward as OF
from helpers import make_conv_weights


@pytest.mark.parametrize("k,stride,act", [(1, 1, "mish"), (3, 1, "mish"), (3, 2, "leaky"), (3, 1, "leaky"), (1, 1, None)])
def test_conv_block_equals_torch_nn_modules(k, stride, act):
    rng = np.random.default_rng(k * 10 + stride)
    cin, cout = 8, 12
    cw = make_conv_weights(rng, cout, cin, k, bn=act is not None)
    x = rng.standard_normal((2, 10, 14, cin)).astype(np.float32)
    layers = []
    if stride == 2:
        layers.append(torch.nn.ZeroPad2d((1, 0, 1, 0)))                      # left, right, top, bottom
    conv = torch.nn.Conv2d(cin, cout, k, stride=stride, padding=0 if stride == 2 else k // 2, bias=act is None)
    conv.weight.data = torch.from_numpy(cw.w.copy())
    if act is None:
        conv.bias.data = torch.from_numpy(cw.bias.copy())
    layers.append(conv)
    if act is not None:
        bn = torch.nn.BatchNorm2d(cout, eps=1e-3)
        beta, gamma, mean, var = (torch.from_numpy(r.copy()) for r in cw.bn)  # Darknet row order
        bn.weight.data, bn.bias.data, bn.running_mean.data, bn.running_var.data = gamma, beta, mean, var
        layers.append(bn)
        layers.append(torch.nn.Mish() if act == "mish" else torch.nn.LeakyReLU(0.1))
    net = torch.nn.Sequential(*layers).eval()
    with torch.no_grad():
        want = net(torch.from_numpy(x).permute(0, 3, 1, 2)).permute(0, 2, 3, 1).numpy()
    got = OF.conv_block(x, cw, k, stride, act)
    assert got.shape == want.shape
    assert np.abs(got - want).max() < 2e-5 * max(1.0, np.abs(want).max())


def _naive_class_nms(boxes, scores, iou_thr, score_thr, per_class, total):
    """Textbook per-class greedy NMS + global top-k for ONE image; returns [(score, box, class)]."""
    def iou(a, b):
        ay0, ay1, ax0, ax1 = min(a[0], a[2]), max(a[0], a[2]), min(a[1], a[3]), max(a[1], a[3])
        by0, by1, bx0, bx1 = min(b[0], b[2]), max(b[0], b[2]), min(b[1], b[3]), max(b[1], b[3])
        aa, ab = np.float32(ay1 - ay0) * np.float32(ax1 - ax0), np.float32(by1 - by0) * np.float32(bx1 - bx0)
        if aa <= 0 or ab <= 0:
            return np.float32(0)
        ih = max(np.float32(0), np.float32(min(ay1, by1) - max(ay0, by0)))
        iw = max(np.float32(0), np.float32(min(ax1, bx1) - max(ax0, bx0)))
        inter = np.float32(ih * iw)
        return np.float32(inter / np.float32(np.float32(aa + ab) - inter))
    out = []
    for c in range(scores.shape[1]):
        cand = sorted((i for i in range(len(boxes)) if scores[i, c] > score_thr), key=lambda i: (-float(scores[i, c]), i))
        kept = []
        for i in cand:
            if len(kept) >= per_class:
                break
            if all(not (iou(boxes[i], boxes[j]) > iou_thr) for j in kept):
                kept.append(i)
        out += [(float(scores[i, c]), i, c) for i in kept]
    out.sort(key=lambda r: (-r[0], r[1], r[2]))
    return out[:total]


def test_combined_nms_equals_naive_definition():
    hyp = pytest.importorskip("hypothesis")
    from hypothesis import given, settings, strategies as st

    @settings(max_examples=60, deadline=None)
    @given(st.integers(0, 2 ** 31 - 1), st.integers(1, 40), st.integers(1, 4), st.sampled_from([3, 100]))
    def run(seed, nbox, ncls, total):
        rng = np.random.default_rng(seed)
        ctr = rng.random((nbox, 2)).astype(np.float32)
        wh = (rng.random((nbox, 2)) * 0.5).astype(np.float32)
        boxes = np.concatenate([ctr - wh / 2, ctr + wh / 2], axis=1).astype(np.float32)     # may leave [0,1]: clipping
        scores = rng.choice(np.linspace(0, 1, 21, dtype=np.float32), size=(nbox, ncls))      # many exact ties and 0.3s
        b, s, c, v, idx = OD.combined_nms(boxes[None], scores[None], 100, total, 0.413, 0.3)
        want = _naive_class_nms(boxes, scores, np.float32(0.413), np.float32(0.3), 100, total)
        assert v[0] == len(want)
        for k, (sc, i, cl) in enumerate(want):
            assert idx[0, k] == i and c[0, k] == cl and s[0, k] == np.float32(sc)
            assert np.array_equal(b[0, k], np.clip(boxes[i], 0, 1))
        assert not s[0, len(want):].any() and (idx[0, len(want):] == -1).all()

    run()
