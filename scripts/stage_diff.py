"""Where does the stage kernel (convs 2..7 fused) differ from the separate kernels?  Debug aid (GPU)."""
import sys, os
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "yolo-v4-tf.keras_amd")); sys.path.insert(0, ROOT)
from yolo4hip import weights as W
from yolo4hip.config import make_config
from yolo4hip.engine import Engine
from yolo4hip.plan import build_plan

size, n, dtype = int(sys.argv[1]), int(sys.argv[2]), sys.argv[3]
plan = build_plan(size, 3)
ws = W.synth_weights(plan, 8)
imgs = W.synth_images(n, size, 8)
eng = Engine(3, make_config(size), max_batch=n, dtype=dtype)
eng.load_weight_blob(W.flatten(ws))
eng.forward_heads(imgs)
ref = eng.conv_output(7, n)
eng.set_stage_fusion(True)
eng.forward_heads(imgs)
got = eng.conv_output(7, n)
bad = got != ref
print("shape", ref.shape, "mismatching elements", int(bad.sum()), "of", bad.size, "max abs diff", float(np.abs(got - ref).max()))
if bad.any():
    b = bad.any(axis=3)
    print("bad pixels per image:", b.reshape(n, -1).sum(axis=1))
    ys, xs = np.nonzero(b.any(axis=0))
    print("bad rows (mod 16) histogram:", np.bincount(ys % 16, minlength=16))
    print("bad cols (mod 16) histogram:", np.bincount(xs % 16, minlength=16))
    print("bad channels:", np.nonzero(bad.any(axis=(0, 1, 2)))[0][:64])
    print("first bad:", np.argwhere(bad)[:5], got[bad][:5], ref[bad][:5])
