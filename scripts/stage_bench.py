"""Times the stage kernel alone (per-op HIP events over whole predicts): usage stage_bench.py [steps]
Env YOLO4HIP_LIB selects a variant build (scripts/build_variant.sh)."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "yolo-v4-tf.keras_amd")); sys.path.insert(0, ROOT)
import torch
from yolo4hip import weights as W
from yolo4hip.config import make_config
from yolo4hip.engine import Engine
from yolo4hip.plan import build_plan
steps = int(sys.argv[1]) if len(sys.argv) > 1 else 10
size, n = 608, 32
plan = build_plan(size, 80)
eng = Engine(80, make_config(size), max_batch=n, dtype="bf16")
eng.load_weight_blob(W.flatten(W.synth_weights(plan, 0)))
imgs = torch.from_numpy(W.synth_images(n, size, 0)).to(eng.device)
eng.set_stem_fusion(True); eng.set_chain_fusion(True); eng.set_stage_fusion(True)
outs = eng.alloc_outputs(n)
for _ in range(3):
    eng.predict_device(imgs, outs)
torch.cuda.synchronize()
eng.timing_begin(steps, coarse=False)
for _ in range(steps):
    eng.predict_device(imgs, outs)
ops, _ = eng.timing_end()
d = dict(ops)
print(os.environ.get("YOLO4HIP_LIB", "default"), "stage kernel %.1f us   stem %.1f us   c8 %.1f us" % (d["c2+3"] * 1e3, d["c0"] * 1e3, d["c8"] * 1e3))
