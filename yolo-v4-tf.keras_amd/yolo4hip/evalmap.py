"""VOC-style mean average precision over the text files `Yolov4.export_gt` / `export_prediction` write
(SURVEY.md f-2): the reference's only accuracy tool, `Yolov4.eval_map` (reference models.py:182-507) with
`voc_ap` / `read_txt_to_list` (reference utils.py:311-356, :469-475).

Protocol reproduced (it is the VOC2012 devkit's, as the reference's docstring says):
  * ground truth files `<id>.txt`: `<class> <left> <top> <right> <bottom>` per line; prediction files `<id>.txt`:
    `<class> <confidence> <left> <top> <right> <bottom>`; every ground-truth file must have a prediction file
    (AssertionError otherwise, models.py:195-196); a prediction file without ground truth is only reported;
  * per class (classes = those present in the ground truth, sorted): detections of all images sorted by descending
    confidence (stable: file order, then line order, break ties); each is matched to the same-class ground-truth box of
    its image with the highest IoU, IoU computed with the devkit's inclusive-pixel `+ 1` convention
    (models.py:303-311); it is a true positive when that IoU >= 0.5 and the box is not yet used, else a false positive
    (duplicates included);
  * AP = area under the precision envelope over the recall steps (`voc_ap`); mAP = mean over the ground-truth classes.
Outputs kept: the prints (class list and counts, `fp/tp/recall/prec` totals, `xx.xx% = <class> AP`, `mAP = xx.xx%`),
`<output>/output.txt` (header line + mAP line, as the reference writes it), and the temp JSON files the reference
leaves behind (`<id>_ground_truth.json` with the final `used` flags, `<class>_dr.json`).  Not kept: the matplotlib
windows (`plt.show()` per class, `draw_plot_func` bar charts) -- display only.  Superset: the function also RETURNS
`{"mAP": ..., "ap": {class: AP}, "tp": {...}, "fp": {...}, "n_gt": {...}}` (the reference returns None).
"""
import glob
import json
import os


def read_txt_to_list(path):
    """Lines of a text file without surrounding whitespace (reference utils.py:469-475)."""
    with open(path) as fh:
        return [line.strip() for line in fh.readlines()]


def voc_ap(rec, prec):
    """(ap, mrec, mpre) from cumulative recall / precision lists (reference utils.py:311-356, the VOC2012 matlab
    code): sentinels (0, 0) and (1, 0) are added, precision is made monotonically non-increasing from the right, and
    the area is summed over the points where recall changes.  Like the reference it edits the lists it is given."""
    rec.insert(0, 0.0); rec.append(1.0)
    prec.insert(0, 0.0); prec.append(0.0)
    mrec, mpre = rec[:], prec[:]
    for i in reversed(range(len(mpre) - 1)):
        if mpre[i + 1] > mpre[i]:
            mpre[i] = mpre[i + 1]
    ap = 0.0
    for i in range(1, len(mrec)):
        step = mrec[i] - mrec[i - 1]
        if mrec[i] != mrec[i - 1]:
            ap += step * mpre[i]
    return ap, mrec, mpre


def _iou_inclusive(bb, gt):
    """IoU with the devkit's inclusive pixel convention (width = right - left + 1); <= 0 when the boxes do not overlap."""
    iw = min(bb[2], gt[2]) - max(bb[0], gt[0]) + 1
    ih = min(bb[3], gt[3]) - max(bb[1], gt[1]) + 1
    if iw <= 0 or ih <= 0:
        return -1.0
    union = (bb[2] - bb[0] + 1) * (bb[3] - bb[1] + 1) + (gt[2] - gt[0] + 1) * (gt[3] - gt[1] + 1) - iw * ih
    return iw * ih / union


def eval_map(gt_folder_path, pred_folder_path, temp_json_folder_path, output_files_path, min_overlap=0.5, verbose=True):
    say = print if verbose else (lambda *a, **k: None)
    gt_files = sorted(glob.glob(gt_folder_path + '/*.txt'))
    assert len(gt_files) > 0, 'no ground truth file'
    # ---- ground truth: per image a list of boxes, per class the object and image counts
    gt_boxes, n_gt, n_img = {}, {}, {}
    for path in gt_files:
        file_id = os.path.basename(os.path.normpath(path.split(".txt", 1)[0]))
        pred_path = os.path.join(pred_folder_path, file_id + ".txt")
        assert os.path.exists(pred_path), "Error. File not found: {}\n".format(pred_path)
        boxes, seen = [], set()
        for line in read_txt_to_list(path):
            cls, left, top, right, bottom = line.split()
            boxes.append({"class_name": cls, "bbox": " ".join((left, top, right, bottom)), "used": False})
            n_gt[cls] = n_gt.get(cls, 0) + 1
            if cls not in seen:
                seen.add(cls)
                n_img[cls] = n_img.get(cls, 0) + 1
        gt_boxes[file_id] = boxes
    gt_classes = sorted(n_gt)
    say(gt_classes, n_gt)
    # ---- predictions, grouped per ground-truth class in descending confidence
    dr_files = sorted(glob.glob(os.path.join(pred_folder_path, '*.txt')))
    per_class = {cls: [] for cls in gt_classes}
    det_count = {}
    for path in dr_files:
        file_id = os.path.basename(os.path.normpath(path.split(".txt", 1)[0]))
        if not os.path.exists(os.path.join(gt_folder_path, file_id + ".txt")):
            say(f"Error. File not found: {os.path.join(gt_folder_path, file_id + '.txt')}\n")
        for line in read_txt_to_list(path):
            parts = line.split()
            if len(parts) != 6:
                say(f"Error: File {path} in the wrong format.\n Expected: <class_name> <confidence> <left> <top> <right> "
                    f"<bottom>\n Received: {line} \n")
                continue
            cls, conf, left, top, right, bottom = parts
            det_count[cls] = det_count.get(cls, 0) + 1
            if cls in per_class:
                per_class[cls].append({"confidence": conf, "file_id": file_id, "bbox": " ".join((left, top, right, bottom))})
    for cls in gt_classes:
        per_class[cls].sort(key=lambda d: float(d["confidence"]), reverse=True)       # stable, like the reference
        with open(os.path.join(temp_json_folder_path, cls + "_dr.json"), "w") as fh:
            json.dump(per_class[cls], fh)
    # ---- AP per class
    aps, tps, fps = {}, {}, {}
    with open(os.path.join(output_files_path, "output.txt"), "w") as out:
        out.write("# AP and precision/recall per class\n")
        for cls in gt_classes:
            dets = per_class[cls]
            tp, fp = [0] * len(dets), [0] * len(dets)
            for k, det in enumerate(dets):
                bb = [float(x) for x in det["bbox"].split()]
                best, match = -1.0, None
                for obj in gt_boxes.get(det["file_id"], ()):
                    if obj["class_name"] != cls:
                        continue
                    ov = _iou_inclusive(bb, [float(x) for x in obj["bbox"].split()])
                    if ov > best:
                        best, match = ov, obj
                if best >= min_overlap and not match["used"]:
                    tp[k] = 1
                    match["used"] = True
                else:
                    fp[k] = 1                                  # low overlap, or a second detection of a used box
            n_fp = n_tp = 0
            for k in range(len(dets)):                         # running totals
                n_fp += fp[k]; fp[k] = n_fp
                n_tp += tp[k]; tp[k] = n_tp
            say('fp ', n_fp)
            say('tp ', n_tp)
            say('recall ', n_tp)
            say('prec ', n_tp)
            rec = [tp[k] / n_gt[cls] for k in range(len(dets))]
            prec = [tp[k] / (fp[k] + tp[k]) for k in range(len(dets))]
            ap, _mrec, _mpre = voc_ap(rec[:], prec[:])
            aps[cls], tps[cls], fps[cls] = ap, n_tp, n_fp
            say("{0:.2f}%".format(ap * 100) + " = " + cls + " AP ")
        m_ap = sum(aps.values()) / len(gt_classes)
        text = "mAP = {0:.2f}%".format(m_ap * 100)
        out.write("\n# mAP of all classes\n")
        out.write(text + "\n")
        say(text)
    for file_id, boxes in gt_boxes.items():                    # what the reference's temp folder holds when it is done
        with open(os.path.join(temp_json_folder_path, file_id + "_ground_truth.json"), "w") as fh:
            json.dump(boxes, fh)
    for cls in det_count:                                      # classes predicted but absent from the ground truth
        tps.setdefault(cls, 0)
    return {"mAP": m_ap, "ap": aps, "tp": tps, "fp": fps, "n_gt": n_gt, "n_images": n_img, "n_det": det_count}
