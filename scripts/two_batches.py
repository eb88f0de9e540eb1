"""Experiment: S full batches in flight -- S engines (own activation workspace, shared nothing) on S HIP streams, steps issued
round-robin -- so that one batch's partial last rounds / 32-workgroup NMS / small 19^2 layers overlap the other's kernels.
usage: two_batches.py <S> [tiles.json]"""
import sys, os, time, json
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "yolo-v4-tf.keras_amd")); sys.path.insert(0, ROOT)
import torch
from yolo4hip import weights as W
from yolo4hip.config import make_config
from yolo4hip.engine import Engine
from yolo4hip.plan import build_plan
S = int(sys.argv[1]); B = 32; steps = 40
plan = build_plan(608, 80); flat = W.flatten(W.synth_weights(plan, 0))
engs = [Engine(80, make_config(608), max_batch=B, dtype="bf16") for _ in range(S)]
streams = [torch.cuda.Stream() for _ in range(S)]
imgs = [torch.from_numpy(W.synth_images(B, 608, 0, first_index=i * B)).cuda() for i in range(S)]
outs = [e.alloc_outputs_flat(B) for e in engs]
hosts = [torch.empty(o[0].numel(), dtype=torch.int32).pin_memory() for o in outs]
tiles = None
for e, x, o in zip(engs, imgs, outs):
    e.load_weight_blob(flat)
    e.set_stem_fusion(True); e.set_chain_fusion(True); e.set_stage_fusion(True); e.set_res_fusion(True)
    e.predict_device(x, o[1])
    if tiles is None:
        tiles = e.autotune(B, reps=3); st = e.stage_fusion_active(); rm = e.res_fusion_mask()
    else:
        e.set_tiles(tiles); e.set_stage_fusion(st); e.set_res_fusion_mask(rm)
torch.cuda.synchronize()
def step(i):
    k = i % S
    with torch.cuda.stream(streams[k]):
        engs[k].predict_device(imgs[k], outs[k][1])
        hosts[k].copy_(outs[k][0], non_blocking=True)
for i in range(2 * S): step(i)
torch.cuda.synchronize()
res = []
for rep in range(5):
    t0 = time.perf_counter()
    for i in range(steps): step(i)
    torch.cuda.synchronize()
    res.append(time.perf_counter() - t0)
dt = sorted(res)[2]
print(f"batches in flight {S}: {B*steps/dt:.1f} img/s, {dt/steps*1e3:.3f} ms per step of {B} images (median of 5 x {steps} steps)")
