"""Per-op times of the 152^2 / 76^2 stage ops with the residual-block kernels forced on / off (GPU).  usage: res_bench.py"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "yolo-v4-tf.keras_amd")); sys.path.insert(0, ROOT)
import json, torch
from yolo4hip import weights as W
from yolo4hip.config import make_config
from yolo4hip.engine import Engine
from yolo4hip.plan import build_plan
size, n = 608, 32
eng = Engine(80, make_config(size), max_batch=n, dtype="bf16")
eng.load_weight_blob(W.flatten(W.synth_weights(build_plan(size, 80), 0)))
imgs = torch.from_numpy(W.synth_images(n, size, 0)).to(eng.device)
eng.set_stem_fusion(True); eng.set_chain_fusion(True); eng.set_stage_fusion(True)
tiles = json.load(open(sys.argv[1]))["tiles"] if len(sys.argv) > 1 else None
if tiles: eng.set_tiles(tiles)
outs = eng.alloc_outputs(n)
def run(mask, steps=10):
    eng.set_res_fusion_mask(mask)
    for _ in range(3): eng.predict_device(imgs, outs)
    torch.cuda.synchronize()
    eng.timing_begin(steps, coarse=False)
    for _ in range(steps): eng.predict_device(imgs, outs)
    ops, _ = eng.timing_end()
    d = dict(ops)
    def idx(k):
        try: return int(k[1:].split("+")[0])
        except ValueError: return -1
    s152 = sum(v for k, v in d.items() if 11 <= idx(k) <= 16)
    s76 = sum(v for k, v in d.items() if 20 <= idx(k) <= 37)
    print(f"mask {mask}: 152^2 ops 11..16 {s152*1e3:.0f} us   76^2 ops 20..37 {s76*1e3:.0f} us   c20 {d.get('c20',0)*1e3:.1f}  c22 {d.get('c22',0)*1e3:.1f} c21 {d.get('c21',0)*1e3:.1f} total {sum(d.values()):.3f} ms")
for m in (0, 1, 2, 3, 0, 3):
    run(m)
