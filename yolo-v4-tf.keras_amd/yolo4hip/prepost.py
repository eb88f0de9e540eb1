"""Host pre/post-processing of the predict() path.

The reference uses OpenCV (`cv2.imread`, `cv2.resize`, `cv2.rectangle`, `cv2.putText`), which is not
installed on the build or GPU machines; these are PIL/NumPy counterparts with the same contracts:

  imread_rgb          reference models.py:126   cv2.imread(path)[:, :, ::-1]  -> RGB uint8 [h,w,3]
  resize_bilinear     reference models.py:96    cv2.resize(img, (W,H)) (INTER_LINEAR, plain stretch)
  preprocess_img      reference models.py:95-98 resize, then img / 255. (float64 in [0,1])
  get_detection_data  reference utils.py:56-78  element 0 of the 4 NMS outputs -> pandas DataFrame
  draw_bbox           reference utils.py:88-118 rectangles + labels (+ plt.imshow when show_img)

cv2's uint8 INTER_LINEAR is fixed point (11-bit coefficients); `resize_bilinear` restates that scheme so
results agree with cv2 to within 1 LSB, but without cv2 here that cannot be verified, so the parity
contract of the hot path starts at the float [N,H,W,3] tensor (BASELINE.json: "same 608x608 inputs").
"""
import numpy as np


def imread_rgb(path):
    from PIL import Image
    try:
        with Image.open(path) as im:
            return np.asarray(im.convert("RGB"), dtype=np.uint8)
    except FileNotFoundError:
        # cv2.imread returns None for a missing file and the reference then fails on `None[:, :, ::-1]`
        raise TypeError("'NoneType' object is not subscriptable") from None


def _linear_coeffs(dst, src):
    scale = src / dst
    f = (np.arange(dst, dtype=np.float64) + 0.5) * scale - 0.5
    s = np.floor(f).astype(np.int64)
    f = (f - s).astype(np.float32)
    lo = s < 0
    f[lo] = 0.0; s[lo] = 0
    hi = s >= src - 1
    f[hi] = 0.0; s[hi] = src - 1
    s1 = np.minimum(s + 1, src - 1)
    a1 = np.rint(f * 2048.0).astype(np.int64)
    a0 = np.rint((1.0 - f) * 2048.0).astype(np.int64)
    return s, s1, a0, a1


def resize_bilinear(img, size_wh):
    """uint8 [h,w,c] -> uint8 [H,W,c]; half-pixel-centre bilinear stretch, OpenCV-style fixed point."""
    img = np.asarray(img)
    W, H = int(size_wh[0]), int(size_wh[1])
    h, w = img.shape[:2]
    if img.dtype != np.uint8:
        return _resize_bilinear_float(img, W, H)
    if (h, w) == (H, W):
        return img.copy()
    x0, x1, ax0, ax1 = _linear_coeffs(W, w)
    y0, y1, ay0, ay1 = _linear_coeffs(H, h)
    src = img.astype(np.int64)
    rows = src[:, x0] * ax0[None, :, None] + src[:, x1] * ax1[None, :, None]          # horizontal pass, x2048
    top, bot = rows[y0], rows[y1]
    out = (((ay0[:, None, None] * (top >> 4)) >> 16) + ((ay1[:, None, None] * (bot >> 4)) >> 16) + 2) >> 2
    return np.clip(out, 0, 255).astype(np.uint8)


def _resize_bilinear_float(img, W, H):
    h, w = img.shape[:2]
    x0, x1, ax0, ax1 = _linear_coeffs(W, w)
    y0, y1, ay0, ay1 = _linear_coeffs(H, h)
    src = img.astype(np.float64)
    fx = (ax1 / 2048.0)[None, :, None]
    rows = src[:, x0] * (1 - fx) + src[:, x1] * fx
    fy = (ay1 / 2048.0)[:, None, None]
    return (rows[y0] * (1 - fy) + rows[y1] * fy).astype(img.dtype)


def preprocess_img(img, img_size):
    """reference models.py:95-98: stretch-resize to img_size[:2] (no letterbox), scale to [0,1] float64."""
    img = resize_bilinear(img, (img_size[1], img_size[0]))
    return img / 255.


def get_detection_data(img, model_outputs, class_names):
    """The DataFrame `predict` returns (contract of reference utils.py:56-78, pinned by tests/test_ref_fixtures.py against
    the reference's own function): batch element 0 only, the first `valid` rows; x scaled by the raw width and y by the raw
    height, truncated to int64; columns x1 y1 x2 y2 class_name score w h; prints the box count."""
    import pandas as pd
    boxes_all, scores_all, classes_all, valid = model_outputs
    n = valid[0]
    raw_h, raw_w = img.shape[:2]
    table = pd.DataFrame(boxes_all[0][:n], columns=['x1', 'y1', 'x2', 'y2'])
    for cols, extent in ((['x1', 'x2'], raw_w), (['y1', 'y2'], raw_h)):
        table[cols] = (table[cols] * extent).astype('int64')
    table['class_name'] = np.array(class_names)[classes_all[0][:n].astype('int64')]
    table['score'] = scores_all[0][:n]
    table['w'] = table['x2'] - table['x1']
    table['h'] = table['y2'] - table['y1']
    print(f'# of bboxes: {n}')
    return table


def draw_bbox(img, detections, cmap, random_color=True, figsize=(10, 10), show_img=True, show_text=True):
    """reference utils.py:88-118 with PIL drawing instead of cv2; returns the annotated RGB uint8 array."""
    from PIL import Image, ImageDraw
    img = np.array(img)
    scale = max(img.shape[0:2]) / 416
    line_width = max(int(2 * scale), 1)
    canvas = Image.fromarray(np.ascontiguousarray(img))
    draw = ImageDraw.Draw(canvas)
    for _, row in detections.iterrows():
        x1, y1, x2, y2, cls, score, w, h = row.values
        color = list(np.random.random(size=3) * 255) if random_color else cmap[cls]
        color = tuple(int(c) for c in color)
        xa, xb = sorted((int(x1), int(x2)))
        ya, yb = sorted((int(y1), int(y2)))
        draw.rectangle([xa, ya, xb, yb], outline=color, width=line_width)
        if show_text:
            text = f'{cls} {score:.2f}'
            l, t, r, b = draw.textbbox((0, 0), text)
            tw, th = r - l, b - t
            draw.rectangle([xa - line_width // 2, ya - th - 2, xa + tw, ya], fill=color)
            draw.text((xa, ya - th - 2), text, fill=(255, 255, 255))
    out = np.asarray(canvas)
    if show_img:
        import matplotlib.pyplot as plt
        plt.figure(figsize=figsize)
        plt.imshow(out)
        plt.show()
    return out
