"""Per-op times with the committed r02 tiles (env YOLO4HIP_LIB selects a variant library)."""
import os, sys, json
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
eng.set_stem_fusion(True); eng.set_chain_fusion(True); eng.set_stage_fusion(True)
eng.set_tiles(json.load(open(os.path.join(ROOT, "profiles/r02/tiles.json")))["tiles"])
outs = eng.alloc_outputs(n)
for _ in range(3): eng.predict_device(imgs, outs)
torch.cuda.synchronize()
eng.timing_begin(10, coarse=False)
for _ in range(10): eng.predict_device(imgs, outs)
d = dict(eng.timing_end()[0])
sel = ["c42", "c44", "c63", "c73", "c81", "c88", "c100", "c104", "c21"]
print(os.environ.get("YOLO4HIP_LIB", "default"), " ".join(f"{k}={d[k]*1e3:.1f}" for k in sel), "total=%.3f" % sum(d.values()))
