#!/usr/bin/env python3
"""Per-workgroup timeline of one conv_halo2_kernel launch (variant library built with -DH2_TRACE): when each workgroup started / ended
(100 MHz realtime), on which XCC / CU, and at which shader clock it ran.  env as h2_one.py."""
import ctypes as C, os, sys, runpy, collections
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
runpy.run_path(os.path.join(ROOT, "scripts", "h2_one.py"), run_name="__main__")
sys.path.insert(0, os.path.join(ROOT, "yolo-v4-tf.keras_amd"))
from yolo4hip import ext
import numpy as np
lib = ext.load()
buf = np.zeros((4096, 4), dtype=np.uint64)
lib.y4_h2_trace_blocks.argtypes = [C.c_void_p]
assert lib.y4_h2_trace_blocks(buf.ctypes.data) == 0
b = buf.astype(np.int64)
n = int(np.count_nonzero(b[:, 1]))
b = b[:n]
t0 = b[:, 0].min()
st, en = (b[:, 0] - t0) / 100.0, (b[:, 1] - t0) / 100.0
hw = b[:, 2] & 0xffffffff
xcc = b[:, 2] >> 32
cu = (hw >> 8) & 0xf; sh = (hw >> 12) & 1; se = (hw >> 13) & 0x7
cuid = xcc * 1000 + se * 100 + sh * 10 + cu
ghz = b[:, 3] / np.maximum(1, b[:, 1] - b[:, 0]) * 0.1
print(f"{n} workgroups, kernel span {en.max():.1f} us; distinct CUs {len(set(cuid.tolist()))}")
order = np.argsort(st)
q = [0, n // 4, n // 2, 3 * n // 4, n - 1]
print("start times (us) quantiles:", [round(float(np.sort(st)[i]), 1) for i in q])
print("end   times (us) quantiles:", [round(float(np.sort(en)[i]), 1) for i in q])
print("duration (us) mean %.1f min %.1f max %.1f; clock GHz mean %.3f min %.3f max %.3f" % ((en - st).mean(), (en - st).min(), (en - st).max(), ghz.mean(), ghz.min(), ghz.max()))
per = collections.defaultdict(list)
for i in range(n):
    per[int(cuid[i])].append((float(st[i]), float(en[i]), i, float(ghz[i])))
cnt = collections.Counter(len(v) for v in per.values())
print("workgroups per CU:", dict(cnt))
for k in list(sorted(per))[:3]:
    print("CU", k, " ".join(f"[{a:.1f}-{e:.1f} wg{i} {g:.2f}GHz]" for a, e, i, g in sorted(per[k])))
