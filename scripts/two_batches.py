"""Experiment: S full batches in flight -- an engine and S-1 siblings (own activation workspace, shared packed weights) on S HIP
streams, steps issued round-robin (yolo4hip.engine.InFlight) -- so that one batch's partial last rounds / 32-workgroup NMS /
small 19^2 layers overlap the other's kernels.
usage: two_batches.py <S> [solo|pair [passes]]     solo: tiles tuned on one engine alone (y4_autotune), pair: tuned with two batches in
                                          flight as objective (y4_autotune_pair; needs S >= 2)"""
import sys, os, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "yolo-v4-tf.keras_amd")); sys.path.insert(0, ROOT)
import torch
from yolo4hip import weights as W
from yolo4hip.config import make_config
from yolo4hip.engine import Engine, InFlight
from yolo4hip.plan import build_plan
S = int(sys.argv[1]); mode = sys.argv[2] if len(sys.argv) > 2 else "solo"; B = 32; steps = 40
passes = int(sys.argv[3]) if len(sys.argv) > 3 else 15      # pair mode: which decisions use the two-stream objective (bits 0..3)
plan = build_plan(608, 80)
eng = Engine(80, make_config(608), max_batch=B, dtype="bf16", alias_workspace=True)
eng.load_weight_blob(W.flatten(W.synth_weights(plan, 0)))
eng.set_stem_fusion(True); eng.set_chain_fusion(True); eng.set_stage_fusion(True); eng.set_res_fusion(True)
imgs = [torch.from_numpy(W.synth_images(B, 608, 0, first_index=i * B)).cuda() for i in range(S)]
eng.predict_device(imgs[0])
if mode == "solo":
    eng.autotune(B, reps=3)
fl = InFlight(eng, S)
if mode == "pair":
    fl.engines[1].predict_device(imgs[1])
    fl.autotune(B, reps=3, passes=passes)
outs = [e.alloc_outputs_flat(B) for e in fl.engines]
hosts = [torch.empty(o[0].numel(), dtype=torch.int32).pin_memory() for o in outs]
torch.cuda.synchronize()
def step(i):
    k = i % S
    fl.submit(imgs[k], outs[k][1], hosts[k], outs[k][0])
for i in range(2 * S): step(i)
torch.cuda.synchronize()
res = []
for rep in range(5):
    t0 = time.perf_counter()
    for i in range(steps): step(i)
    torch.cuda.synchronize()
    res.append(time.perf_counter() - t0)
dt = sorted(res)[2]
print(f"batches in flight {S} ({mode}-tuned{'' if mode == 'solo' else ' passes ' + str(passes)}), res mask {eng.res_fusion_mask()}: {B*steps/dt:.1f} img/s, {dt/steps*1e3:.3f} ms per step of {B} images (median of 5 x {steps} steps)")
