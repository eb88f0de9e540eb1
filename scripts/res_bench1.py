import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "yolo-v4-tf.keras_amd")); sys.path.insert(0, ROOT)
import torch
from yolo4hip import weights as W
from yolo4hip.config import make_config
from yolo4hip.engine import Engine
from yolo4hip.plan import build_plan
size, n = 608, 32
eng = Engine(80, make_config(size), max_batch=n, dtype="bf16")
eng.load_weight_blob(W.flatten(W.synth_weights(build_plan(size, 80), 0)))
imgs = torch.from_numpy(W.synth_images(n, size, 0)).to(eng.device)
eng.set_stem_fusion(True); eng.set_chain_fusion(True); eng.set_stage_fusion(True); eng.set_res_fusion(True)
outs = eng.alloc_outputs(n)
for _ in range(3): eng.predict_device(imgs, outs)
torch.cuda.synchronize()
eng.timing_begin(8, coarse=False)
for _ in range(8): eng.predict_device(imgs, outs)
d = dict(eng.timing_end()[0])
print(os.environ.get("YOLO4HIP_LIB", "default"), "resblock128 c22 %.1f us  resblock64 c11 %.1f us" % (d["c22"] * 1e3, d["c11"] * 1e3))
