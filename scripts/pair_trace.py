"""Life trace of an LDS-pair head kernel (conv_igemm_kernel<..., PAIR = true> on the 192x256 tile; a Y4_TRACE variant build):

    bash scripts/build_variant.sh trp conv_igemm_bf16_fused "-DY4_TRACE=1 -DY4_TRACE_BM=192 -DY4_TRACE_NST=2 -DY4_TRACE_PAIR=1"
    YOLO4HIP_LIB=scratch/libyolo4hip_trp.so python scripts/pair_trace.py [last_conv]

Runs the 608/80/bf16 batch-32 model on the shipped schedule up to conv `last_conv` (default 43: the pair 42 -> 43, a 3x3 256->256 + 1x1 at
38^2) -- the last pair-head launch is the traced one -- and prints for workgroup 8, per wave, the shader-clock offsets (from the first
wave's "head K loop done") of: first K-tile in (not reliable: s_memtime results of the kernel's first instructions are not waited for),
head K loop done, head tile in LDS (its epilogue + barrier), tail K loop done, tail epilogue done (its stores issued), the head tile's
own store issued; with the wall time from s_memrealtime."""
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

last = int(sys.argv[1]) if len(sys.argv) > 1 else 43
size, n = 608, 32
eng = Engine(80, make_config(size), max_batch=n, dtype="bf16", alias_workspace=True)
eng.load_weight_blob(W.flatten(W.synth_weights(build_plan(size, 80), 0)))
eng.ensure_schedule(tune=False)
imgs = torch.from_numpy(W.synth_images(n, size, 0)).to(eng.device)
for _ in range(3): eng.forward_until_device(imgs, last)
torch.cuda.synchronize()
lib = ext.load()
life = (C.c_ulonglong * 64)()
lib.y4_trace_read_life_fused.restype = C.c_int
assert lib.y4_trace_read_life_fused(life) == 0, "not a Y4_TRACE build"
L = np.array(life[:], dtype=np.int64).reshape(8, 8)
cols = [2, 3, 7, 0, 1, 4]
names = ["first tile in", "head K loop done", "tile in LDS", "tail K loop done", "tail epilogue done", "all stored"]
l0 = L[:, 3].min()
print("pair head ending at conv %d, workgroup 8 (shader cycles from the first wave's entry):" % last)
for w in range(8):
    r = L[w, cols] - l0
    print("  wave %d: " % w + "  ".join("%s %6d" % (names[i], r[i]) for i in range(6)) + "   | dt: " +
          " ".join("%6d" % (r[i + 1] - r[i]) for i in range(5)) + "   (%.2f us wall)" % ((L[w, 6] - L[w, 5]) / 100.0))
