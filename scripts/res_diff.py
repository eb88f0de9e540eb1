"""Where does the residual-block kernel differ from the separate kernels?  Debug aid (GPU): res_diff.py size n dtype"""
import sys, os
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "yolo-v4-tf.keras_amd")); sys.path.insert(0, ROOT)
from yolo4hip import weights as W
from yolo4hip.config import make_config
from yolo4hip.engine import Engine
from yolo4hip.plan import build_plan
size, n, dtype = int(sys.argv[1]), int(sys.argv[2]), sys.argv[3]
eng = Engine(3, make_config(size), max_batch=n, dtype=dtype)
eng.load_weight_blob(W.flatten(W.synth_weights(build_plan(size, 3), 8)))
imgs = W.synth_images(n, size, 8)
eng.forward_heads(imgs)
taps = (12, 14, 21, 23, 35)
ref = {i: eng.conv_output(i, n) for i in taps}
print("res runs:", eng.set_res_fusion(True))
eng.forward_heads(imgs)
for i in taps:
    got = eng.conv_output(i, n)
    bad = got != ref[i]
    print("conv", i, "shape", got.shape, "mismatching", int(bad.sum()), "of", bad.size, "max abs diff", float(np.abs(got - ref[i]).max()))
    if bad.any():
        b = bad.any(axis=3)
        ys, xs = np.nonzero(b.any(axis=0))
        print("  bad px per image", b.reshape(n, -1).sum(axis=1), "rows%16", np.bincount(ys % 16, minlength=16), "cols%16", np.bincount(xs % 16, minlength=16))
        print("  bad channels", np.nonzero(bad.any(axis=(0, 1, 2)))[0][:40])
        break
