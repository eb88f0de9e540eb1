import sys, os, time
ROOT = "/root/repo" if os.path.isdir("/root/repo/yolo-v4-tf.keras_amd") else os.getcwd()
sys.path.insert(0, os.path.join(ROOT, "yolo-v4-tf.keras_amd")); sys.path.insert(0, ROOT)
import torch
from yolo4hip import weights as W
from yolo4hip.config import make_config
from yolo4hip.engine import Engine, InFlight
from yolo4hip.plan import build_plan
size, ncls, n = 608, 80, 32
eng = Engine(ncls, make_config(size), max_batch=n, dtype="bf16", alias_workspace=True)
eng.load_weight_blob(W.flatten(W.synth_weights(build_plan(size, ncls), 0)))
eng.set_stem_fusion(True); eng.set_chain_fusion(True); eng.set_stage_fusion(True); eng.set_res_fusion(True)
eng.apply_schedule(eng.shipped_schedule())
imgs = torch.from_numpy(W.synth_images(n, size, seed=0)).to(eng.device)
fl = InFlight(eng, 2)
outs = [e.alloc_outputs(n) for e in fl.engines]
for _ in range(6): fl.submit(imgs, outs[_ % 2])
fl.synchronize()
t0 = time.perf_counter(); host = []
for i in range(200):
    a = time.perf_counter(); fl.submit(imgs, outs[i % 2]); host.append(time.perf_counter() - a)
fl.synchronize(); t1 = time.perf_counter()
host.sort()
print(f"200 steps: wall {1e3*(t1-t0)/200:.3f} ms/step; host time inside submit: median {1e3*host[100]:.3f} ms, p90 {1e3*host[180]:.3f} ms, min {1e3*host[0]:.3f}")
