"""GPU parity of decode + class-aware NMS (y4_set_heads + y4_decode_nms through the C ABI) against the
oracle (oracle/decode_nms.py <- reference custom_layers.py:201-298 + tf.image.combined_non_max_suppression).
Decisions (valid counts, kept indices, classes) must be identical; boxes/scores within 1e-5 / 1e-6
(float32 expf/division may differ from NumPy's by an ulp)."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu


def _engine(size, ncls, n):
    from yolo4hip import weights as W
    from yolo4hip.config import make_config
    from yolo4hip.engine import Engine
    from yolo4hip.plan import build_plan
    cfg = make_config(size)
    eng = Engine(ncls, cfg, max_batch=n, dtype="bf16")
    eng.adopt_packed()          # decode/NMS do not read weights
    return cfg, eng


def _compare(eng, cfg, heads, size, ncls, iou=-1.0, score=-1.0):
    from oracle import decode_nms as OD
    n = eng.set_heads(heads)
    got = [o.cpu().numpy() for o in eng.decode_nms_device(n, None, iou, score)]
    ref = OD.inference_from_heads(heads, ncls, cfg["anchors"], cfg["xyscale"], size,
                                  iou_threshold=cfg["iou_threshold"] if iou < 0 else iou,
                                  score_threshold=cfg["score_threshold"] if score < 0 else score)
    assert np.array_equal(got[3], ref[3]), (got[3], ref[3])
    assert np.array_equal(got[4], ref[4])
    assert np.array_equal(got[2], ref[2])
    assert np.abs(got[0] - ref[0]).max() < 1e-5
    assert np.abs(got[1] - ref[1]).max() < 1e-6
    return got, ref


def _random_heads(rng, n, size, ncls, obj_bias, cls_bias, gain=1.5):
    heads = []
    nf = 5 + ncls
    for s in (8, 16, 32):
        g = size // s
        h = (rng.standard_normal((n, g, g, 3, nf)) * gain).astype(np.float32)
        h[..., 2:4] *= 0.3
        h[..., 4] += obj_bias
        h[..., 5:] += cls_bias
        heads.append(h.reshape(n, g, g, 3 * nf))
    return heads


def test_zero_logits_closed_form():
    """All-zero logits: sigmoid = 0.5 everywhere -> score 0.25 < 0.3 -> no detections; with the threshold
    lowered every box passes with score exactly 0.25 and xy = (grid + 0.5) * stride, wh = anchor."""
    size, ncls = 96, 2
    cfg, eng = _engine(size, ncls, 1)
    heads = [np.zeros((1, size // s, size // s, 3 * (5 + ncls)), np.float32) for s in (8, 16, 32)]
    got, _ = _compare(eng, cfg, heads, size, ncls)
    assert got[3][0] == 0 and np.all(got[0] == 0) and np.all(got[4] == -1)
    got, _ = _compare(eng, cfg, heads, size, ncls, score=0.2)
    assert got[3][0] == 100 and np.all(got[1][0] == 0.25)
    # first kept = box 0, class 0 (ties: box index asc, class asc): cell (0,0), anchor (12,16), stride 8
    x1, y1, x2, y2 = got[0][0, 0]
    assert got[4][0, 0] == 0 and got[2][0, 0] == 0
    assert abs(x1 - 0.0) < 1e-7 and abs(x2 - (4 + 6) / size) < 1e-6      # x1 = (4-6)/96 clipped to 0
    assert abs(y1 - 0.0) < 1e-7 and abs(y2 - (4 + 8) / size) < 1e-6
    eng.close()


@pytest.mark.parametrize("size,ncls,n,obj_bias,cls_bias", [
    (416, 80, 2, -3.0, -3.0),      # typical: O(10^2..10^3) candidates
    (416, 3, 3, -1.0, 0.0),        # few classes, thousands of candidates in one class (> SORT_CAP chunks)
    (160, 6, 2, -6.0, -2.0),       # sparse: fewer than 100 detections, some images may have none
    (608, 80, 1, -2.0, -2.5),
])
def test_random_logits_vs_oracle(size, ncls, n, obj_bias, cls_bias):
    cfg, eng = _engine(size, ncls, n)
    rng = np.random.default_rng(size + ncls)
    heads = _random_heads(rng, n, size, ncls, obj_bias, cls_bias)
    _compare(eng, cfg, heads, size, ncls)
    eng.close()


def test_dense_overlapping_boxes_many_chunks():
    """Everything passes the score threshold (>> SORT_CAP candidates per image, heavy overlap): exercises
    the radix-select chunk loop and long suppression chains."""
    size, ncls, n = 224, 3, 2
    cfg, eng = _engine(size, ncls, n)
    rng = np.random.default_rng(11)
    heads = _random_heads(rng, n, size, ncls, 3.0, 2.0, gain=0.7)
    got, ref = _compare(eng, cfg, heads, size, ncls)
    assert got[3].min() > 0
    eng.close()


def test_custom_thresholds_like_predict_nonms():
    size, ncls, n = 160, 4, 2
    cfg, eng = _engine(size, ncls, n)
    rng = np.random.default_rng(2)
    heads = _random_heads(rng, n, size, ncls, -2.0, -1.0)
    _compare(eng, cfg, heads, size, ncls, iou=0.413, score=0.1)     # reference models.py:516 defaults
    _compare(eng, cfg, heads, size, ncls, iou=0.9, score=0.5)
    _compare(eng, cfg, heads, size, ncls, iou=0.0, score=0.05)
    eng.close()


def _heads_with(size, ncls, n, boxes):
    """Logits that are hugely negative everywhere except the listed (image, scale, gy, gx, anchor, class, obj_logit, cls_logit,
    txywh) cells: exactly those boxes are candidates."""
    nf = 5 + ncls
    heads = [np.full((n, size // s, size // s, 3, nf), -20.0, np.float32) for s in (8, 16, 32)]
    for h in heads:
        h[..., :4] = 0.0
    for (b, sc, gy, gx, a, c, lo, lc, t) in boxes:
        heads[sc][b, gy, gx, a, :4] = t
        heads[sc][b, gy, gx, a, 4] = lo
        heads[sc][b, gy, gx, a, 5 + c] = lc
    return [h.reshape(n, h.shape[1], h.shape[2], 3 * nf) for h in heads]


def test_empty_single_and_mixed_images():
    """Edge cases of tf.image.combined_non_max_suppression's padded outputs: an image with no candidate at all (valid 0, all
    zeros), one with exactly one, and both in ONE batch next to a crowded image -- rows are independent."""
    size, ncls, n = 160, 5, 3
    cfg, eng = _engine(size, ncls, n)
    rng = np.random.default_rng(5)
    crowded = _random_heads(rng, 1, size, ncls, 1.0, 0.0)
    one = _heads_with(size, ncls, 1, [(0, 1, 3, 4, 2, 3, 6.0, 6.0, (0.2, -0.1, 0.3, 0.1))])
    none = _heads_with(size, ncls, 1, [])
    heads = [np.concatenate([none[s], one[s], crowded[s]], axis=0) for s in range(3)]
    got, ref = _compare(eng, cfg, heads, size, ncls)
    assert list(got[3][:2]) == [0, 1] and got[3][2] > 1
    assert np.all(got[0][0] == 0) and np.all(got[1][0] == 0) and np.all(got[2][0] == 0) and np.all(got[4][0] == -1)
    assert got[2][1, 0] == 3 and np.all(got[1][1, 1:] == 0)
    eng.close()


def test_identical_boxes_of_different_classes_all_survive():
    """Class-aware suppression: the SAME box scored for every class is kept once per class (IoU 1 only suppresses within a
    class), in descending score order; two identical boxes of ONE class collapse to the better one."""
    size, ncls, n = 96, 4, 1
    cfg, eng = _engine(size, ncls, n)
    t = (0.0, 0.0, 0.2, 0.2)
    boxes = [(0, 0, 5, 5, 1, c, 8.0, 2.0 + c, t) for c in range(ncls)]
    heads = _heads_with(size, ncls, n, boxes)
    # all four class logits sit in the same cell/anchor: one box, four classes
    got, ref = _compare(eng, cfg, heads, size, ncls)
    assert got[3][0] == ncls and list(got[2][0, :ncls]) == [3, 2, 1, 0]
    assert np.all(got[0][0, :ncls] == got[0][0, 0])
    # the neighbouring anchor slot with the same decoded box is a different box index of the same class 0 -> suppressed
    eng.close()


def test_more_than_the_cap_of_disjoint_boxes():
    """> max_total (100) non-overlapping boxes of one class above the threshold: exactly the 100 best survive, in score order,
    ties by box index (the oracle's stable order)."""
    size, ncls, n = 416, 2, 1
    cfg, eng = _engine(size, ncls, n)
    boxes = []
    g = size // 8
    k = 0
    for gy in range(0, g, 4):
        for gx in range(0, g, 4):
            boxes.append((0, 0, gy, gx, 0, 1, 6.0, 3.0 + 0.01 * (k % 37), (0.0, 0.0, -0.5, -0.5)))
            k += 1
    assert len(boxes) > 150
    heads = _heads_with(size, ncls, n, boxes)
    got, ref = _compare(eng, cfg, heads, size, ncls)
    assert got[3][0] == 100 and np.all(np.diff(got[1][0]) <= 0) and np.all(got[2][0] == 1)
    eng.close()


def test_one_score_bin_holds_more_than_a_chunk_under_a_few_better_keys():
    """three candidates in the top score bins and ~2 000 inside ONE 2^-9-wide bin below them (more than the 1 024 keys of
    the first NMS chunk): the histogram pivot's suffix would be those three keys alone -- the kernel must fall through to the exact
    radix select and still return the oracle's 100 boxes (score desc, ties by box index then class)."""
    size, ncls, n = 608, 2, 1
    cfg, eng = _engine(size, ncls, n)
    g = size // 8
    boxes = [(0, 0, 1, 1 + 9 * i, 0, 0, 8.0, 8.0, (0.0, 0.0, -1.0, -1.0)) for i in range(3)]        # three confident boxes
    k = 0
    for gy in range(4, g, 2):
        for gx in range(0, g, 2):
            if k >= 2000:
                break
            # sigmoid(1.0) * sigmoid(1.0 + tiny) ~ 0.534: the scores of all of these share one histogram bin (score bits >> 14)
            boxes.append((0, 0, gy, gx, k % 3, k % 2, 1.0, 1.0 + 1e-4 * (k % 11), (0.0, 0.0, -1.5, -1.5)))
            k += 1
    assert k > 1100
    heads = _heads_with(size, ncls, n, boxes)
    got, ref = _compare(eng, cfg, heads, size, ncls)
    assert got[3][0] == 100 and np.all(np.diff(got[1][0]) <= 0)
    assert got[1][0, 0] > 0.99 and got[1][0, 3] < 0.6
    eng.close()
