"""`eval_map` / `voc_ap` (reference models.py:182-507, utils.py:311-356) against hand-computed known answers.
The protocol is the VOC2012 devkit's: detections sorted by confidence, greedy match to the best-IoU ground-truth box of
the same class and image (IoU >= 0.5, inclusive-pixel +1 convention), duplicates are false positives, AP = area under
the monotone precision envelope."""
import json
import os
import sys

import numpy as np
import pytest

sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "yolo-v4-tf.keras_amd"))
from yolo4hip.evalmap import eval_map, read_txt_to_list, voc_ap   # noqa: E402


def test_voc_ap_known_answers():
    # recall/precision after each detection for tp = [1,0,1,0] with 3 ground-truth boxes
    rec, prec = [1 / 3, 1 / 3, 2 / 3, 2 / 3], [1.0, 0.5, 2 / 3, 0.5]
    ap, mrec, mpre = voc_ap(rec[:], prec[:])
    assert mrec == [0.0, 1 / 3, 1 / 3, 2 / 3, 2 / 3, 1.0]
    assert mpre == [1.0, 1.0, 2 / 3, 2 / 3, 0.5, 0.0]                  # monotone envelope, sentinel 0 at recall 1
    assert ap == pytest.approx(1 / 3 * 1.0 + 1 / 3 * (2 / 3) + 1 / 3 * 0.0)
    assert voc_ap([1.0], [1.0])[0] == pytest.approx(1.0)               # one perfect detection
    assert voc_ap([], [])[0] == 0.0                                    # no detections at all
    assert voc_ap([0.0, 0.0], [0.0, 0.0])[0] == 0.0                    # only false positives
    # the lists handed in are edited in place, like the reference's (callers pass copies)
    r, p = [0.5], [1.0]
    voc_ap(r, p)
    assert r == [0.0, 0.5, 1.0] and p == [0.0, 1.0, 0.0]


def _write(folder, name, lines):
    with open(os.path.join(folder, name + ".txt"), "w") as fh:
        fh.write("".join(l + "\n" for l in lines))


def test_eval_map_hand_built_case(tmp_path, capsys):
    gt, pr, tmp, out = (str(tmp_path / d) for d in ("gt", "pred", "tmp", "out"))
    for d in (gt, pr, tmp, out):
        os.makedirs(d)
    # image a: two cats, one dog; image b: one cat
    _write(gt, "a", ["cat 10 10 50 50", "cat 100 100 150 150", "dog 200 200 260 260"])
    _write(gt, "b", ["cat 20 20 80 80"])
    # cat detections by confidence: 0.9 a hit (TP), 0.8 b miss (FP), 0.7 b hit (TP), 0.6 a duplicate of the first (FP)
    # dog: 0.95 a hit (IoU 0.70 with the +1 convention) -> AP 1;  bird: not in the ground truth -> ignored for mAP
    _write(pr, "a", ["cat 0.9 12 12 50 50", "cat 0.6 10 10 48 48", "dog 0.95 205 205 265 265", "bird 0.5 1 1 5 5"])
    _write(pr, "b", ["cat 0.8 200 200 240 240", "cat 0.7 22 18 82 78"])
    _write(pr, "c", ["cat 0.99 0 0 10 10"])             # prediction without ground truth: reported, still counted (FP)
    res = eval_map(gt, pr, tmp, out)
    text = capsys.readouterr().out
    # cat: order 0.99(c, FP) 0.9(TP) 0.8(FP) 0.7(TP) 0.6(FP); 3 ground-truth cats
    #   tp = [0,1,1,2,2], fp = [1,1,2,2,3] -> rec = [0,1/3,1/3,2/3,2/3], prec = [0,1/2,1/3,1/2,2/5]
    #   envelope at recall 1/3: 1/2, at 2/3: 1/2, at 1: 0  ->  AP = 1/3*1/2 + 1/3*1/2 = 1/3
    assert res["ap"]["cat"] == pytest.approx(1 / 3)
    assert res["ap"]["dog"] == pytest.approx(1.0)
    assert res["mAP"] == pytest.approx((1 / 3 + 1.0) / 2)
    assert res["tp"] == {"cat": 2, "dog": 1, "bird": 0} and res["fp"] == {"cat": 3, "dog": 0}
    assert res["n_gt"] == {"cat": 3, "dog": 1} and res["n_det"]["bird"] == 1
    assert "['cat', 'dog']" in text and "33.33% = cat AP" in text and "100.00% = dog AP" in text and "mAP = 66.67%" in text
    assert "File not found" in text                       # the prediction file without ground truth
    assert open(os.path.join(out, "output.txt")).read() == \
        "# AP and precision/recall per class\n\n# mAP of all classes\nmAP = 66.67%\n"
    used = json.load(open(os.path.join(tmp, "a_ground_truth.json")))
    assert [b["used"] for b in used] == [True, False, True]
    dr = json.load(open(os.path.join(tmp, "cat_dr.json")))
    assert [d["confidence"] for d in dr] == ["0.99", "0.9", "0.8", "0.7", "0.6"] and dr[0]["file_id"] == "c"
    assert read_txt_to_list(os.path.join(gt, "b.txt")) == ["cat 20 20 80 80"]


def test_eval_map_requires_prediction_files(tmp_path):
    gt, pr, tmp, out = (str(tmp_path / d) for d in ("gt", "pred", "tmp", "out"))
    for d in (gt, pr, tmp, out):
        os.makedirs(d)
    with pytest.raises(AssertionError, match="no ground truth file"):
        eval_map(gt, pr, tmp, out)
    _write(gt, "a", ["cat 1 1 5 5"])
    with pytest.raises(AssertionError, match="File not found"):
        eval_map(gt, pr, tmp, out)


def test_iou_threshold_edge_is_inclusive(tmp_path):
    """IoU exactly 0.5 counts (`ovmax >= min_overlap`, models.py:315); boxes are inclusive-pixel rectangles."""
    gt, pr, tmp, out = (str(tmp_path / d) for d in ("gt", "pred", "tmp", "out"))
    for d in (gt, pr, tmp, out):
        os.makedirs(d)
    _write(gt, "a", ["x 0 0 9 9"])                        # 10 x 10 = 100 pixels
    _write(pr, "a", ["x 0.5 0 0 9 4"])                    # 10 x 5 = 50 pixels inside it: IoU = 50 / 100
    assert eval_map(gt, pr, tmp, out, verbose=False)["tp"]["x"] == 1
    _write(pr, "a", ["x 0.5 0 0 8 4"])                    # 45 / 100 < 0.5
    assert eval_map(gt, pr, tmp, out, verbose=False)["tp"]["x"] == 0
