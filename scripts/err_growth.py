"""Per-layer error growth: HIP fp32 vs oracle fp32, and both vs the oracle in fp64 (error budgeting)."""
import sys, os
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "yolo-v4-tf.keras_amd")); sys.path.insert(0, ROOT)
import numpy as np, torch
from yolo4hip import weights as W
from yolo4hip.config import make_config
from yolo4hip.engine import Engine
from yolo4hip.plan import build_plan
from oracle import forward as OF
size, ncls, n = int(sys.argv[1]), int(sys.argv[2]), 1
dtype = sys.argv[3] if len(sys.argv) > 3 else "f32"
plan = build_plan(size, ncls); ws = W.synth_weights(plan, 0); imgs = W.synth_images(n, size, 0)
idxs = list(range(0, 110, 3)) + [107, 108]
h32, t32 = OF.yolo_model_forward(imgs, ws, ncls, collect=idxs)
h64, t64 = OF.yolo_model_forward(imgs, ws, ncls, dtype=torch.float64, collect=idxs)
eng = Engine(ncls, make_config(size), max_batch=n, dtype=dtype); eng.load_weight_blob(W.flatten(ws))
hg = eng.forward_heads(imgs)
for i in sorted(set(idxs)):
    g = eng.conv_output(i, n)
    a, b = t32.get(("add", i), t32[i]), t64.get(("add", i), t64[i])
    if i in (78, 85): a = a.repeat(2, 1).repeat(2, 2); b = b.repeat(2, 1).repeat(2, 2)
    print(f"c{i:3d} gpu-vs-o32 {np.abs(g-a).max():.2e}  gpu-vs-o64 {np.abs(g-b).max():.2e}  o32-vs-o64 {np.abs(a-b).max():.2e}  |x|max {np.abs(b).max():.1f}")
for k in range(3):
    print(f"head{k} gpu-vs-o32 {np.abs(hg[k]-h32[k]).max():.2e} gpu-vs-o64 {np.abs(hg[k]-h64[k]).max():.2e} o32-vs-o64 {np.abs(h32[k]-h64[k]).max():.2e}")
