"""In-kernel phase trace of stem_down_kernel (an SD_TRACE variant build, see stem_down.hip):

    bash scripts/build_variant.sh sdtr stem_down "-DSD_TRACE=1"
    YOLO4HIP_LIB=scratch/libyolo4hip_sdtr.so python scripts/stem_trace.py

Runs the 608/80/bf16 batch-32 model and prints, for workgroup 8 and its output rows 5..8, per wave the cycles between the
trace points (waves w, w+4, w+8, w+12 share a SIMD)."""
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
eng.set_stem_fusion(True); eng.set_chain_fusion(True); eng.set_stage_fusion(True)
outs = eng.alloc_outputs(n)
for _ in range(3): eng.predict_device(imgs, outs)
torch.cuda.synchronize()
lib = ext.load()
buf = (C.c_ulonglong * (4 * 16 * 8))()
lib.y4_sd_trace_read.restype = C.c_int
assert lib.y4_sd_trace_read(buf) == 0, "not an SD_TRACE build"
raw = np.array(buf[:], dtype=np.int64).reshape(4, 16, 8)[:, :, :6]
names = ["start", "stem", "bar1", "conv", "epi", "bar2"]
t0 = raw[0, :, 0].min()
print("cycles per output row (wave 0, start -> start):", [int(raw[k + 1, 0, 0] - raw[k, 0, 0]) for k in range(3)])
for k in range(4):
    print("row", 5 + k)
    for w in range(16):
        r = raw[k, w] - t0
        print("  wave %2d: start %7d | dt: " % (w, r[0]) + " ".join("%s %5d" % (names[i + 1], r[i + 1] - r[i]) for i in range(5)))
