"""Experiment: split the batch over S engines on S HIP streams (independent sub-batches) so that one sub-batch's
memory-bound epilogues/tails overlap the other's MFMA phases."""
import sys, os, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "yolo-v4-tf.keras_amd")); sys.path.insert(0, ROOT)
import torch
from yolo4hip import weights as W
from yolo4hip.config import make_config
from yolo4hip.engine import Engine
from yolo4hip.plan import build_plan
S = int(sys.argv[1]); B = 32; steps = 20
plan = build_plan(608, 80); flat = W.flatten(W.synth_weights(plan, 0))
per = B // S
engs = [Engine(80, make_config(608), max_batch=per, dtype="bf16") for _ in range(S)]
streams = [torch.cuda.Stream() for _ in range(S)]
imgs = [torch.from_numpy(W.synth_images(per, 608, 0, first_index=i * per)).cuda() for i in range(S)]
outs = [e.alloc_outputs(per) for e in engs]
for e in engs:
    e.load_weight_blob(flat)
for e, x, o in zip(engs, imgs, outs):
    e.predict_device(x, o); e.autotune(per)
torch.cuda.synchronize()
def step():
    for e, x, o, st in zip(engs, imgs, outs, streams):
        with torch.cuda.stream(st):
            e.predict_device(x, o)
for _ in range(3): step()
torch.cuda.synchronize()
t0 = time.perf_counter()
for _ in range(steps): step()
torch.cuda.synchronize()
dt = time.perf_counter() - t0
print(f"streams {S}: {B*steps/dt:.1f} img/s, {dt/steps*1e3:.3f} ms per {B} images")
