"""ORACLE (test infrastructure, not product code): NumPy restatement of the reference's head decode and
class-aware NMS.

PARITY: `get_boxes` / `yolov4_head` / the `nms` wrapper are PINNED to the reference's own code -- `tests/golden/
make_ref_fixtures.py` executes /root/reference/custom_layers.py:201-298 unmodified under a torch-backed TensorFlow
stand-in and `tests/test_ref_fixtures.py` compares (decode to one ulp, NMS outputs identical).  UNPINNED:
`tf.image.combined_non_max_suppression` itself, which lives in un-vendored TensorFlow (only version evidence:
`tf.__version__ == '2.2.0'` in notebook/Inference.ipynb cell 0); its published CPU algorithm
(tensorflow/core/kernels/non_max_suppression_op.cc, BatchedNonMaxSuppressionOp) is restated below from its documented
behaviour, anchored on the reference's only call site `custom_layers.py:290-297`, and cross-checked against two other
independent writings of the same semantics (the stand-in's, and the brute force of tests/test_oracle_crosscheck.py).

Only `tests/`, `__graft_entry__.smoke()` and `bench.py`'s `cpu_baseline` leg may import this module.

Follows `/root/reference/custom_layers.py`:
  get_boxes    :221-258   reshape [N,g,g,3,5+C]; sigmoid(xy,obj,cls); grid (x=col, y=row);
                          xy = ((sig*xyscale) - 0.5*(xyscale-1) + grid) * stride; wh = exp(t)*anchor;
                          x1y1 = xy - wh/2, x2y2 = xy + wh/2
  yolov4_head  :201-218   the three scales (grid sizes generalised from 52/26/13 to img//stride)
  nms          :261-298   flatten each scale in (row, col, anchor) order, concat scales 0,1,2;
                          scores = obj * cls; boxes / input_shape[0];
                          combined_non_max_suppression(100/class, 100 total, iou 0.413, score 0.3),
                          defaults pad_per_class=False, clip_boxes=True
and `/root/reference/config.py:4-6,14-16`, `/root/reference/models.py:29` (anchors.reshape(3,3,2)).

Tie order among EQUAL scores is heap/std::sort dependent in TensorFlow, i.e. not contractual; this
restatement (and the HIP kernels) define it as (score desc, box index asc, class asc).
"""
import numpy as np

F32 = np.float32


def sigmoid(x):
    # NOTE: exact float32 1/(1+exp(-x)).  TensorFlow's CPU kernel for tf.sigmoid is Eigen's clamped rational `logistic`
    # approximation, which differs from this by a few ulp: scores within ~1e-7 relative of the 0.3 cut can fall on the other
    # side in TensorFlow.  Identity of kept indices is therefore defined against THIS restatement (LABNOTES.md section 2).
    x = np.asarray(x, dtype=F32)
    return (F32(1.0) / (F32(1.0) + np.exp(-x))).astype(F32)


def get_boxes(pred, anchors, classes, grid_size, strides, xyscale):
    """custom_layers.py:221-258.  pred [N,g,g,3*(5+C)] float32 -> (x1y1x2y2 [N,g,g,3,4] in input
    pixels, obj [N,g,g,3,1], cls [N,g,g,3,C], xywh [N,g,g,3,4])."""
    pred = np.asarray(pred, dtype=F32)
    n = pred.shape[0]
    pred = pred.reshape(n, grid_size, grid_size, 3, 5 + classes)
    box_xy, box_wh = pred[..., 0:2], pred[..., 2:4]
    obj_prob, class_prob = pred[..., 4:5], pred[..., 5:]
    box_xy = sigmoid(box_xy)
    obj_prob = sigmoid(obj_prob)
    class_prob = sigmoid(class_prob)
    pred_box_xywh = np.concatenate((box_xy, box_wh), axis=-1)
    gx, gy = np.meshgrid(np.arange(grid_size), np.arange(grid_size))       # gx[i,j]=j, gy[i,j]=i
    grid = np.stack([gx, gy], axis=-1)[:, :, None, :].astype(F32)          # (g,g,1,2): (x=col, y=row)
    box_xy = ((box_xy * F32(xyscale)) - F32(0.5 * (xyscale - 1)) + grid) * F32(strides)
    box_wh = np.exp(box_wh).astype(F32) * np.asarray(anchors, dtype=F32)
    box_x1y1 = box_xy - box_wh / F32(2)
    box_x2y2 = box_xy + box_wh / F32(2)
    return np.concatenate([box_x1y1, box_x2y2], axis=-1).astype(F32), obj_prob, class_prob, pred_box_xywh


def yolov4_head(outputs, classes, anchors, xyscale, strides=(8, 16, 32)):
    """custom_layers.py:201-218 with grid_size = side of each output (== img_size // stride)."""
    anchors = np.asarray(anchors, dtype=F32).reshape(3, 3, 2)
    res = []
    for s in range(3):
        g = outputs[s].shape[1]
        res.extend(get_boxes(outputs[s], anchors[s], classes, g, strides[s], xyscale[s]))
    return res


def flatten_for_nms(head_outputs, input_size, num_class):
    """custom_layers.py:269-284: -> boxes [N,nbox,4] normalised (x1,y1,x2,y2), scores [N,nbox,C]."""
    n = head_outputs[0].shape[0]
    boxes = np.zeros((n, 0, 4), F32)
    conf = np.zeros((n, 0, 1), F32)
    cls = np.zeros((n, 0, num_class), F32)
    for i in range(0, len(head_outputs), 4):
        boxes = np.concatenate([boxes, head_outputs[i].reshape(n, -1, 4)], axis=1)
        conf = np.concatenate([conf, head_outputs[i + 1].reshape(n, -1, 1)], axis=1)
        cls = np.concatenate([cls, head_outputs[i + 2].reshape(n, -1, num_class)], axis=1)
    scores = (conf * cls).astype(F32)
    boxes = (boxes / F32(input_size)).astype(F32)
    return boxes, scores


def _iou_one_vs_many(b, others):
    """TensorFlow's IOU(): corners min/max-normalised, 0 if either area <= 0; float32 arithmetic."""
    y0 = np.minimum(b[0], b[2]); x0 = np.minimum(b[1], b[3])
    y1 = np.maximum(b[0], b[2]); x1 = np.maximum(b[1], b[3])
    oy0 = np.minimum(others[:, 0], others[:, 2]); ox0 = np.minimum(others[:, 1], others[:, 3])
    oy1 = np.maximum(others[:, 0], others[:, 2]); ox1 = np.maximum(others[:, 1], others[:, 3])
    area_i = F32((y1 - y0) * (x1 - x0))
    area_j = ((oy1 - oy0) * (ox1 - ox0)).astype(F32)
    iy0 = np.maximum(y0, oy0); ix0 = np.maximum(x0, ox0)
    iy1 = np.minimum(y1, oy1); ix1 = np.minimum(x1, ox1)
    inter = (np.maximum(iy1 - iy0, F32(0)) * np.maximum(ix1 - ix0, F32(0))).astype(F32)
    with np.errstate(divide="ignore", invalid="ignore"):
        iou = (inter / (area_i + area_j - inter)).astype(F32)
    iou = np.where((area_i <= 0) | (area_j <= 0), F32(0), iou)
    return iou


def combined_nms(boxes, scores, max_output_size_per_class=100, max_total_size=100,
                 iou_threshold=0.413, score_threshold=0.3, clip_boxes=True):
    """Restatement of tf.image.combined_non_max_suppression for boxes [N,nbox,4] shared across classes
    (q == 1, the reference's `tf.expand_dims(boxes, -2)`) and scores [N,nbox,C].
    Returns nmsed_boxes [N,T,4], nmsed_scores [N,T], nmsed_classes [N,T] (float32), valid [N] int32
    and -- build-only extra -- kept_idx [N,T] int32 (box index n of SURVEY.md §3.4, -1 padded)."""
    boxes = np.asarray(boxes, F32)
    scores = np.asarray(scores, F32)
    n, nbox, ncls = scores.shape
    T = max_total_size
    iou_thr, score_thr = F32(iou_threshold), F32(score_threshold)
    out_b = np.zeros((n, T, 4), F32); out_s = np.zeros((n, T), F32)
    out_c = np.zeros((n, T), F32); out_v = np.zeros((n,), np.int32)
    out_i = np.full((n, T), -1, np.int32)
    for b in range(n):
        result = []   # (score, box_index, class)
        for c in range(ncls):
            sc = scores[b, :, c]
            cand = np.nonzero(sc > score_thr)[0]                     # strict '>'
            if cand.size == 0:
                continue
            order = cand[np.lexsort((cand, -sc[cand].astype(np.float64)))]   # score desc, index asc
            sel = []
            for i in order:
                if len(sel) >= max_output_size_per_class:
                    break
                if sel:
                    iou = _iou_one_vs_many(boxes[b, i], boxes[b, sel])
                    if np.any(iou > iou_thr):                        # suppress when strictly greater
                        continue
                sel.append(int(i))
                result.append((float(sc[i]), int(i), c))
        result.sort(key=lambda r: (-r[0], r[1], r[2]))
        result = result[:T]
        out_v[b] = len(result)
        for k, (s, i, c) in enumerate(result):
            bb = boxes[b, i]
            out_b[b, k] = np.clip(bb, F32(0), F32(1)) if clip_boxes else bb
            out_s[b, k] = s; out_c[b, k] = c; out_i[b, k] = i
    return out_b, out_s, out_c, out_v, out_i


def inference_from_heads(outputs, num_classes, anchors, xyscale, input_size, strides=(8, 16, 32),
                         iou_threshold=0.413, score_threshold=0.3, max_boxes=100):
    """`inference_model` tail (`models.py:68-73`): raw heads -> the 4 NMS outputs (+ kept_idx)."""
    head = yolov4_head(outputs, num_classes, anchors, xyscale, strides)
    boxes, scores = flatten_for_nms(head, input_size, num_classes)
    return combined_nms(boxes, scores, 100, max_boxes, iou_threshold, score_threshold)
