"""In-kernel phase trace of conv_igemm_kernel's K loop (a Y4_TRACE variant build, see conv_igemm_kernel.h):

    bash scripts/build_variant.sh tr19 conv_igemm_bf16 "-DY4_TRACE=1 -DY4_TRACE_BM=192 -DY4_TRACE_NST=2"
    YOLO4HIP_LIB=scratch/libyolo4hip_tr19.so python scripts/trace_read.py [conv] [tile] [--json out.json]

Runs the 608/80/bf16 batch-32 model with the committed tiles, with conv `conv` (default 81: 3x3 256->512 at 38^2) the only
op on tile `tile` (default 19 = 192x256, 2 stages), and prints for workgroup 8, K-tiles 10..13, per wave: the shader-clock
offsets of the trace points and the time spent between them.
  arrive   top of the K-tile iteration (about to wait for the tile's loads and the barrier)
  landed   s_waitcnt vmcnt(0) returned: this wave's own LDS-DMA loads of the tile are in LDS
  barrier  released from the K-tile barrier
  dma      next tile's LDS-DMA instructions issued, staging cursor advanced
  reads    the K-tile's fragment reads (ds_read_b128) issued
  mma      the K-tile's last MFMA issued
and the workgroup's life outside the K loop: kernel entry, staging set-up done, first K-tile's barrier passed, K loop done,
epilogue done (stores issued), with the wall time from s_memrealtime.
The traced K-tiles are ~10 % longer than untraced ones (8 timestamp stores per wave at the end of each)."""
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

args = [a for a in sys.argv[1:] if not a.startswith("--")]
out_json = sys.argv[sys.argv.index("--json") + 1] if "--json" in sys.argv else None
if out_json: args = [a for a in args if a != out_json]
only = int(args[0]) if len(args) > 0 else 81
tile = int(args[1]) if len(args) > 1 else 19
size, n = 608, 32
eng = Engine(80, make_config(size), max_batch=n, dtype="bf16")
eng.load_weight_blob(W.flatten(W.synth_weights(build_plan(size, 80), 0)))
imgs = torch.from_numpy(W.synth_images(n, size, 0)).to(eng.device)
eng.set_stem_fusion(True); eng.set_chain_fusion(True); eng.set_stage_fusion(True)
tiles = json.load(open(os.path.join(ROOT, "profiles/r02/tiles.json")))["tiles"]
for i, t in enumerate(tiles):                      # only conv `only` runs the traced tile
    if t == tile and i != only: tiles[i] = 23
tiles[only] = tile
eng.set_tiles(tiles)
outs = eng.alloc_outputs(n)
for _ in range(3): eng.predict_device(imgs, outs)
torch.cuda.synchronize()
lib = ext.load()
buf = (C.c_ulonglong * 256)()
lib.y4_trace_read.restype = C.c_int
assert lib.y4_trace_read(buf) == 0, "not a Y4_TRACE build"
raw = np.array(buf[:], dtype=np.int64).reshape(4, 8, 8)
names = ["arrive", "landed", "barrier", "dma", "reads", "mma"]
a = raw[:, :, [0, 4, 1, 2, 3, 5]]
t0 = a[0, :, 0].min()
dc, dr = raw[3, 0, 0] - raw[0, 0, 0], raw[3, 0, 7] - raw[0, 0, 7]
clock = dc / dr * 0.1
print(f"conv {only}, tile {tile}: 3 traced K-tiles = {dc} shader cycles = {dr} ticks of the 100 MHz counter -> shader clock "
      f"{clock:.2f} GHz, {dc / 3:.0f} cycles per traced K-tile")
for kt in range(4):
    print("K-tile", 10 + kt)
    for w in range(8):
        r = a[kt, w] - t0
        print("  wave %d: " % w + "  ".join("%s %6d" % (names[i], r[i]) for i in range(6)) +
              "   | dt: " + " ".join("%5d" % (r[i + 1] - r[i]) for i in range(5)))
life = (C.c_ulonglong * 64)()
lib.y4_trace_read_life.restype = C.c_int
assert lib.y4_trace_read_life(life) == 0
L = np.array(life[:], dtype=np.int64).reshape(8, 8)
l0 = L[:, 0].min()
lnames = ["entry", "set-up done", "first tile in", "K loop done", "epilogue done"]
print("workgroup life outside the K loop (shader cycles from the first wave's entry):")
for w in range(8):
    r = L[w, :5] - l0
    print("  wave %d: " % w + "  ".join("%s %6d" % (lnames[i], r[i]) for i in range(5)) +
          "   | dt: " + " ".join("%6d" % (r[i + 1] - r[i]) for i in range(4)) + "   (%.2f us wall)" % ((L[w, 6] - L[w, 5]) / 100.0))
if out_json:
    json.dump({"conv": only, "tile": tile, "workgroup": 8, "k_tiles": [10, 11, 12, 13], "points": names,
               "shader_clock_ghz": round(float(clock), 3), "cycles_per_traced_k_tile": round(float(dc) / 3, 1),
               "cycles": (a - t0).tolist(), "life_points": lnames, "life_cycles": (L[:, :5] - l0).tolist(),
               "life_wall_us": ((L[:, 6] - L[:, 5]) / 100.0).tolist(),
               "note": "cycles[k_tile][wave][point], shader-clock cycles from the first arrival; traced K-tiles carry ~10 % "
                       "extra (timestamp stores)"}, open(out_json, "w"), indent=1)
