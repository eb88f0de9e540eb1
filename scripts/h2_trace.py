#!/usr/bin/env python3
"""Reads the s_memtime stamps of conv_halo2_kernel (variant library built with -DH2_TRACE; YOLO4HIP_LIB points at it) after a few
launches of one tile: per wave of workgroups 0..15 the cycles spent in prologue / each chunk / epilogue.  env as h2_one.py."""
import ctypes as C, os, sys, runpy
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
runpy.run_path(os.path.join(ROOT, "scripts", "h2_one.py"), run_name="__main__")
sys.path.insert(0, os.path.join(ROOT, "yolo-v4-tf.keras_amd"))
from yolo4hip import ext
import numpy as np
lib = ext.load()
buf = np.zeros((16, 4, 16), dtype=np.uint64)
lib.y4_h2_trace_read.argtypes = [C.c_void_p]
rc = lib.y4_h2_trace_read(buf.ctypes.data)
assert rc == 0, rc
b = buf.astype(np.int64)
for wg in (0, 9, 5):
    for wv in range(4):
        t = b[wg, wv]
        nch = int(np.count_nonzero(t[3:12]))
        segs = [t[1] - t[0], t[2] - t[1]] + [t[3 + c] - (t[2] if c == 0 else t[2 + c]) for c in range(nch)]
        print(f"wg {wg:2d} wave {wv}: start {t[0] - b[0,0,0]:7d} | setup {segs[0]:5d} wait0 {segs[1]:5d} | chunks " + " ".join(f"{x:6d}" for x in segs[2:]) + f" | drain {t[12] - t[2 + nch]:5d} | epilogue {t[13] - t[12]:6d} | total {t[13] - t[0]:6d} = {(t[15] - t[14]) / 100.0:.2f} us -> {(t[13] - t[0]) / max(1, t[15] - t[14]) * 0.1:.3f} GHz")
