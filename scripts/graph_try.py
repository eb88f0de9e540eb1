"""Does a HIP graph (torch.cuda.graph around y4_predict) help small batches?  usage: graph_try.py dtype batch"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "yolo-v4-tf.keras_amd")); sys.path.insert(0, ROOT)
import torch
from yolo4hip import weights as W
from yolo4hip.config import make_config
from yolo4hip.engine import Engine
from yolo4hip.plan import build_plan
dtype, n = sys.argv[1], int(sys.argv[2])
mode = sys.argv[3] if len(sys.argv) > 3 else "all"
size = 608
eng = Engine(80, make_config(size), max_batch=n, dtype=dtype)
eng.load_weight_blob(W.flatten(W.synth_weights(build_plan(size, 80), 0)))
imgs = torch.from_numpy(W.synth_images(n, size, 0)).to(eng.device)
if dtype != "f32" and mode in ("all", "fuse"):
    eng.set_stem_fusion(True); eng.set_chain_fusion(True); eng.set_stage_fusion(True)
if mode == "stem": eng.set_stem_fusion(True)
if mode == "chain": eng.set_chain_fusion(True)
if mode == "stage": eng.set_stage_fusion(True)
outs = eng.alloc_outputs(n)
eng.predict_device(imgs, outs)
if mode == "all": eng.autotune(n, reps=3)
for _ in range(5): eng.predict_device(imgs, outs)
torch.cuda.synchronize()
def timeit(fn, k=50):
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(k): fn()
    torch.cuda.synchronize(); return (time.perf_counter() - t0) / k * 1e3
eager = timeit(lambda: eng.predict_device(imgs, outs))
ref = [o.clone() for o in outs]
s = torch.cuda.Stream()
s.wait_stream(torch.cuda.current_stream())
with torch.cuda.stream(s):
    eng.predict_device(imgs, outs)
torch.cuda.current_stream().wait_stream(s)
g = torch.cuda.CUDAGraph()
with torch.cuda.graph(g):
    eng.predict_device(imgs, outs)
for o in outs: o.zero_()
g.replay(); torch.cuda.synchronize()
same = all(torch.equal(a, b) for a, b in zip(ref, outs))
graph = timeit(g.replay)
print(f"{mode} {dtype} batch {n}: eager {eager:.3f} ms/step, graph replay {graph:.3f} ms/step, identical outputs: {same}")
