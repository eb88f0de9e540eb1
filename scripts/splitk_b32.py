"""Experiment: what would split-K ids (and the cold-timed tuner that comes with y4_set_splitk) buy at the HEADLINE shape, batch 32?
Prints images/s one stream and two in flight for the shipped schedule and for a schedule tuned with split-K allowed.  A split launch sums
in another fp32 order, so such a schedule is not in the bit-identical set (LABNOTES.md section 4.7): measured, not shipped."""
import json, os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "yolo-v4-tf.keras_amd")); sys.path.insert(0, ROOT)
import torch
from yolo4hip import weights as W
from yolo4hip.config import make_config
from yolo4hip.engine import Engine, InFlight
from yolo4hip.plan import build_plan

size, n = 608, 32
eng = Engine(80, make_config(size), max_batch=n, dtype="bf16", alias_workspace=True)
eng.load_weight_blob(W.flatten(W.synth_weights(build_plan(size, 80), 0)))
imgs = torch.from_numpy(W.synth_images(n, size, 0)).to(eng.device)
eng.ensure_schedule(tune=False, verbose=True)

def rate(depth, steps=60):
    fl = InFlight(eng, depth)
    outs = [eng.alloc_outputs(n) for _ in range(depth)]
    ims = [imgs] + [imgs.clone() for _ in range(depth - 1)]
    for i in range(2 * depth): fl.submit(ims[i % depth], outs[i % depth])
    torch.cuda.synchronize()
    best = 0
    for _ in range(3):
        t0 = time.perf_counter()
        for i in range(steps): fl.submit(ims[i % depth], outs[i % depth])
        torch.cuda.synchronize()
        best = max(best, n * steps / (time.perf_counter() - t0))
    fl.close()
    return best

print("shipped schedule: one stream %.0f img/s, two in flight %.0f" % (rate(1), rate(2)))
eng.set_splitk(True)
eng.predict_device(imgs)
tiles = eng.autotune(n, reps=3)
eng.set_splitk(False)
split = sorted(set(t for t in tiles if t >= 100))
print("tuned cold with split-K allowed: %d layers on split ids %s, stage %s, res mask %d" % (sum(t >= 100 for t in tiles), split, eng.stage_fusion_active(), eng.res_fusion_mask()))
print("that schedule:     one stream %.0f img/s, two in flight %.0f" % (rate(1), rate(2)))
nosplit = [t if t < 100 else 0 for t in tiles]
