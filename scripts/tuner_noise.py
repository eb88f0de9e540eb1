"""How reproducible is y4_autotune?  N tuning runs of the headline shape in one process: entries that differ between runs,
and the single-stream step time of each run's schedule, alternating.  usage: tuner_noise.py [runs=4]"""
import sys, os, json
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "yolo-v4-tf.keras_amd")); sys.path.insert(0, ROOT)
import torch, time
from yolo4hip import weights as W
from yolo4hip.config import make_config
from yolo4hip.engine import Engine
from yolo4hip.plan import build_plan
NR = int(sys.argv[1]) if len(sys.argv) > 1 else 4
size, ncls, n = 608, 80, 32
eng = Engine(ncls, make_config(size), max_batch=n, dtype="bf16", alias_workspace=True)
eng.load_weight_blob(W.flatten(W.synth_weights(build_plan(size, ncls), 0)))
imgs = torch.from_numpy(W.synth_images(n, size, seed=0)).to(eng.device)
outs = eng.alloc_outputs(n)
def fus(): eng.set_stem_fusion(True); eng.set_chain_fusion(True); eng.set_stage_fusion(True); eng.set_res_fusion(True)
fus(); eng.predict_device(imgs, outs)
ship = eng.shipped_schedule()
runs = []
for r in range(NR):
    fus(); t0 = time.time(); tiles = eng.autotune(n, reps=3); torch.cuda.synchronize()
    runs.append({"tiles": tiles, "stage_fusion": ship["stage_fusion"], "res_fusion_mask": ship["res_fusion_mask"], "name": f"tune{r}", "secs": time.time() - t0})
for a in range(NR):
    print(runs[a]["name"], f"{runs[a]['secs']:.1f} s", "differs from", [(runs[b]["name"], sum(1 for x, y in zip(runs[a]["tiles"], runs[b]["tiles"]) if x != y)) for b in range(NR) if b != a])
def step_ms(s, steps=60):
    fus(); eng.apply_schedule(s)
    for _ in range(5): eng.predict_device(imgs, outs)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(steps): eng.predict_device(imgs, outs)
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / steps
for rnd in range(3):
    print("round", rnd, " | ".join(f"{s.get('name', 'shipped')} {step_ms(s):.4f}" for s in [ship] + runs), flush=True)
