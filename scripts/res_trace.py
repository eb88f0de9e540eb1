"""In-kernel phase trace of resblock_kernel (an RB_TRACE variant build: -DRB_TRACE=128 or -DRB_TRACE=64 selects the channel count):

    bash scripts/build_variant.sh rbtr resblock "-DRB_TRACE=128"
    YOLO4HIP_LIB=scratch/libyolo4hip_rbtr.so python scripts/res_trace.py

Runs the 608/80/bf16 batch-32 model with the shipped schedule and prints, for workgroup 8 of the full-tile launch and its tiles 1..3,
per wave the cycles between the trace points (waves w and w+4 share a SIMD)."""
import ctypes as C
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

size, n = 608, 32
eng = Engine(80, make_config(size), max_batch=n, dtype="bf16")
eng.load_weight_blob(W.flatten(W.synth_weights(build_plan(size, 80), 0)))
imgs = torch.from_numpy(W.synth_images(n, size, 0)).to(eng.device)
eng.ensure_schedule(tune=False, verbose=True)
outs = eng.alloc_outputs(n)
for _ in range(3): eng.predict_device(imgs, outs)
torch.cuda.synchronize()
lib = ext.load()
buf = (C.c_ulonglong * (3 * 8 * 16))()
lib.y4_rb_trace_read.restype = C.c_int
assert lib.y4_rb_trace_read(buf) == 0, "not an RB_TRACE build"
raw = np.array(buf[:], dtype=np.int64).reshape(3, 8, 16)[:, :, :11]
names = ["arrive", "bar0", "1x1", "midbar", "issue", "taps0-2", "taps3-5", "taps6-8", "endbar", "prefetch", "epilogue"]
t0 = raw[0, :, 0].min()
print("cycles per tile (wave 0, arrive -> arrive):", [int(raw[k + 1, 0, 0] - raw[k, 0, 0]) for k in range(2)])
for k in range(3):
    print("tile", 1 + k)
    for w in range(8):
        r = raw[k, w] - t0
        print("  wave %d: arrive %7d | dt: " % (w, r[0]) + " ".join("%s %5d" % (names[i + 1], r[i + 1] - r[i]) for i in range(10)))
