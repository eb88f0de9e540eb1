"""Debug aid: per-op times with a given tile list (negative = chained head), 608x608 batch 32 bf16 (GPU)."""
import sys, os, json
sys.path.insert(0, os.path.join(os.path.dirname(__file__), "..", "yolo-v4-tf.keras_amd"))
import torch
from yolo4hip import weights as W
from yolo4hip.config import make_config
from yolo4hip.engine import Engine
from yolo4hip.plan import build_plan

tiles = json.load(open(sys.argv[1]))["tiles"]
for kv in sys.argv[2:]:
    k, v = kv.split("=")
    tiles[int(k)] = int(v)
size, ncls, n = 608, 80, 32
plan = build_plan(size, ncls)
eng = Engine(ncls, make_config(size), max_batch=n, dtype="bf16")
eng.load_weight_blob(W.flatten(W.synth_weights(plan, 0)))
imgs = torch.from_numpy(W.synth_images(n, size, 0)).to(eng.device)
outs = eng.alloc_outputs(n)
eng.set_stem_fusion(True)
eng.set_chain_fusion(True)
eng.set_tiles(tiles)
for _ in range(3):
    eng.predict_device(imgs, outs)
torch.cuda.synchronize()
eng.timing_begin(10, coarse=False)
for _ in range(10):
    eng.predict_device(imgs, outs)
torch.cuda.synchronize()
ops, _ = eng.timing_end()
print({k: round(v, 4) for k, v in ops if k in ("c20","c21","c22","c23","c41","c42","c43","c44")}, "total", round(sum(v for _, v in ops), 4))
