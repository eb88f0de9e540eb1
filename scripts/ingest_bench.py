"""PCIe-inclusive throughput of the pipelined uint8 ingest path (Engine.predict_stream) next to the device-resident
rate bench.py reports: frames start in ordinary host memory as uint8, results end in host numpy arrays."""
import os, sys, time
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "yolo-v4-tf.keras_amd"))
import numpy as np
import torch
from yolo4hip import weights as W
from yolo4hip.config import make_config
from yolo4hip.engine import Engine
from yolo4hip.plan import build_plan

size, ncls, n = 608, 80, 32
plan = build_plan(size, ncls)
eng = Engine(ncls, make_config(size), max_batch=n, dtype="bf16", alias_workspace=True)
eng.load_weight_blob(W.flatten(W.synth_weights(plan, 0)))
eng.set_stem_fusion(True)
eng.set_chain_fusion(True)
eng.set_stage_fusion(True)
eng.set_res_fusion(True)
if eng.ensure_schedule(tune=False, verbose=True)[0] == "heuristic":       # the shipped schedule of the headline shape, else tune here
    eng.predict_device(torch.from_numpy(W.synth_images(n, size, 0)).to(eng.device))
    eng.autotune(n, reps=3)
IN_FLIGHT = int(sys.argv[1]) if len(sys.argv) > 1 else 2
rng = np.random.default_rng(0)
for (h, w) in ((608, 608), (720, 1280), (1080, 1920)):
    frames = [rng.integers(0, 256, size=(n, h, w, 3), dtype=np.uint8) for _ in range(2)]
    nb = 24
    for _ in eng.predict_stream((frames[i % 2] for i in range(4)), in_flight=IN_FLIGHT):
        pass
    torch.cuda.synchronize()
    best = None
    for _rep in range(2):                     # the first timed pass still pays one-off costs (pinned staging, streams): report the best of two
        t0 = time.perf_counter()
        got = 0
        for res in eng.predict_stream((frames[i % 2] for i in range(nb)), in_flight=IN_FLIGHT):
            got += res[3].shape[0]
        dt = time.perf_counter() - t0
        best = dt if best is None or dt < best else best
    dt = best
    print(f"uint8 {h}x{w} frames, batch {n}, {IN_FLIGHT} in flight: {got / dt:8.1f} images/s PCIe-inclusive ({dt / nb * 1e3:.2f} ms per batch, "
          f"{n * h * w * 3 / 1e6:.1f} MB per batch over PCIe)", flush=True)
# device-resident reference point on the same box
imgs = torch.from_numpy(W.synth_images(n, size, 0)).to(eng.device)
outs = eng.alloc_outputs(n)
for _ in range(3):
    eng.predict_device(imgs, outs)
torch.cuda.synchronize()
t0 = time.perf_counter()
for _ in range(20):
    eng.predict_device(imgs, outs)
torch.cuda.synchronize()
print(f"device-resident float32 inputs: {20 * n / (time.perf_counter() - t0):8.1f} images/s")
