"""Per-op time of selected plain conv layers with a forced tile id each (env YOLO4HIP_LIB selects a variant library).
usage: tile_cmp.py <tile> [<tile> ...]   -- layers: the 3x3 / 1x1 convs of the 38^2 and 19^2 stages that run as plain kernels"""
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
base = json.load(open(os.path.join(ROOT, "profiles/r02/tiles.json")))["tiles"]
sel = [63, 73, 81, 100, 104]
outs = eng.alloc_outputs(n)
def run(tiles):
    eng.set_tiles(tiles)
    for _ in range(3): eng.predict_device(imgs, outs)
    torch.cuda.synchronize()
    eng.timing_begin(10, coarse=False)
    for _ in range(10): eng.predict_device(imgs, outs)
    d = dict(eng.timing_end()[0])
    return {k: d["c%d" % k] * 1e3 for k in sel}, sum(d.values())
r, tot = run(base)
print("base tiles", [base[k] for k in sel], " ".join(f"c{k}={v:.1f}" for k, v in r.items()), "total=%.3f" % tot)
for t in map(int, sys.argv[1:]):
    tiles = list(base)
    for k in sel: tiles[k] = t
    try:
        r, tot = run(tiles)
        print("tile %2d" % t, " ".join(f"c{k}={v:.1f}" for k, v in r.items()), "total=%.3f" % tot)
    except Exception as e:
        print("tile", t, "failed:", str(e)[:100])
