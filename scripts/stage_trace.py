"""In-kernel phase trace of csp_stage_kernel (a CS_TRACE variant build, see csp_stage.hip):

    bash scripts/build_variant.sh cstr csp_stage "-DCS_TRACE=1"
    YOLO4HIP_LIB=scratch/libyolo4hip_cstr.so python scripts/stage_trace.py [--json out.json]

Runs the 608/80/bf16 batch-32 model with the stage kernel on and prints, for workgroup 8 and its tiles 3..6, per wave the
shader-clock offsets of the trace points and the cycles between them (waves w and w+4 share a SIMD)."""
import ctypes as C
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "yolo-v4-tf.keras_amd")); sys.path.insert(0, ROOT)
import numpy as np
import torch
from yolo4hip import weights as W, ext
from yolo4hip.config import make_config
from yolo4hip.engine import Engine
from yolo4hip.plan import build_plan

out_json = sys.argv[sys.argv.index("--json") + 1] if "--json" in sys.argv else None
size, n = 608, 32
eng = Engine(80, make_config(size), max_batch=n, dtype="bf16")
eng.load_weight_blob(W.flatten(W.synth_weights(build_plan(size, 80), 0)))
imgs = torch.from_numpy(W.synth_images(n, size, 0)).to(eng.device)
eng.set_stem_fusion(True); eng.set_chain_fusion(True); eng.set_stage_fusion(True)
outs = eng.alloc_outputs(n)
for _ in range(3): eng.predict_device(imgs, outs)
torch.cuda.synchronize()
lib = ext.load()
buf = (C.c_ulonglong * (4 * 8 * 16))()
lib.y4_cs_trace_read.restype = C.c_int
assert lib.y4_cs_trace_read(buf) == 0, "not a CS_TRACE build"
raw = np.array(buf[:], dtype=np.int64).reshape(4, 8, 16)[:, :, :13]
names = ["arrive", "landed", "m32", "v32", "extra", "c4", "midbar", "dma", "m5", "v5", "c6", "m7", "v7+st"]
t0 = raw[0, :, 0].min()
print("cycles per tile (wave 0, arrive -> arrive):", [int(raw[k + 1, 0, 0] - raw[k, 0, 0]) for k in range(3)])
for k in range(4):
    print("tile", 3 + k)
    for w in range(8):
        r = raw[k, w] - t0
        print("  wave %d: arrive %7d | dt: " % (w, r[0]) + " ".join("%s %5d" % (names[i + 1], r[i + 1] - r[i]) for i in range(12)))
if out_json:
    json.dump({"workgroup": 8, "tiles": [3, 4, 5, 6], "points": names, "cycles": (raw - t0).tolist(),
               "note": "cycles[tile][wave][point], shader-clock cycles from the first arrival; every point is a sched_barrier, so "
                       "the traced kernel is slower than the shipped one"}, open(out_json, "w"), indent=1)
