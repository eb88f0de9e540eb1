"""Which end-to-end detections differ between the HIP fp32 path and the fp32 oracle, and why."""
import sys, os
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "yolo-v4-tf.keras_amd")); sys.path.insert(0, ROOT)
import numpy as np
from yolo4hip import weights as W
from yolo4hip.config import make_config
from yolo4hip.engine import Engine
from yolo4hip.plan import build_plan
from oracle import forward as OF, decode_nms as OD
size, ncls, n = int(sys.argv[1]), int(sys.argv[2]), int(sys.argv[3])
cfg = make_config(size); plan = build_plan(size, ncls); ws = W.synth_weights(plan, 0); imgs = W.synth_images(n, size, 0)
eng = Engine(ncls, cfg, max_batch=n, dtype="f32"); eng.load_weight_blob(W.flatten(ws))
ref_heads = OF.yolo_model_forward(imgs, ws, ncls)
b, s, c, v, k = eng.predict(imgs, with_indices=True)
rb, rs, rc, rv, ri = OD.inference_from_heads(ref_heads, ncls, cfg["anchors"], cfg["xyscale"], size)
for i in range(n):
    for j in range(100):
        if k[i, j] != ri[i, j] or c[i, j] != rc[i, j]:
            print(f"img {i} rank {j}: gpu (idx {k[i,j]}, cls {c[i,j]:.0f}, s {s[i,j]:.6f})  oracle (idx {ri[i,j]}, cls {rc[i,j]:.0f}, s {rs[i,j]:.6f})")
    print("img", i, "valid", v[i], rv[i], "max score diff", np.abs(s[i]-rs[i]).max())
