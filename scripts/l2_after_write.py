"""Does a consumer kernel find its producer's output in L2?  Producer P = 3x3 512->1024 @19^2 (writes X), consumer C = 1x1 1024->512
reading X (the step's c73 -> c74), batch 32, bf16, their tuned tiles.  C's device time (HIP events around C only):
  hot      : C, C, C, ...                      (X read by the previous launch)
  after P  : P, C, P, C, ...                   (X just written by P; same XCD row ranges)
  after Q  : Q, C, Q, C, ...  Q = P writing to ANOTHER buffer (X not written, but P's traffic has gone through L2)
usage: l2_after_write.py"""
import sys, os, ctypes as C
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "yolo-v4-tf.keras_amd")); sys.path.insert(0, ROOT)
import numpy as np, torch
from yolo4hip import ext
lib = ext.load()
dev, td, did = "cuda:0", torch.bfloat16, ext.DTYPE_IDS["bf16"]
N = 32


def conv(k, s, cin, cout, side, x, out, tile):
    cpad, nb = C.c_int32(), C.c_size_t()
    ext.check(lib.y4_packed_conv_bytes(did, cout, cin, k, C.byref(cpad), C.byref(nb)))
    w = torch.randn((cout, cin, k, k), device=dev) * 0.02
    packed = torch.empty(nb.value, dtype=torch.uint8, device=dev)
    ext.check(lib.y4_pack_conv_weights(did, cout, cin, k, ext.ptr(w), ext.ptr(packed), ext.stream_ptr()))
    sc = torch.ones(cpad.value, device=dev); sh = torch.zeros(cpad.value, device=dev)
    d = ext.y4_conv_desc(); d.dtype = did; d.n, d.h, d.w, d.cin = N, side, side, cin
    d.cout, d.ksize, d.stride, d.act = cout, k, s, 1
    d.in_cstride, d.out_cstride = cin, cout
    d.in_ = x.data_ptr(); d.wt = packed.data_ptr(); d.scale = sc.data_ptr(); d.shift = sh.data_ptr(); d.out = out.data_ptr()
    d.tile = tile
    keep = (w, packed, sc, sh)
    return d, keep


side = 19
a = torch.randn((N, side, side, 512), device=dev).to(td)
X = torch.empty((N, side, side, 1024), device=dev, dtype=td)
X2 = torch.empty_like(X)
Y = torch.empty((N, side, side, 512), device=dev, dtype=td)
P, kp = conv(3, 1, 512, 1024, side, a, X, 19)
Q, kq = conv(3, 1, 512, 1024, side, a, X2, 19)
Cc, kc = conv(1, 1, 1024, 512, side, X, Y, 20)
run = lambda d: ext.check(lib.y4_conv2d(C.byref(d), ext.stream_ptr()))
run(P); run(Q); run(Cc); torch.cuda.synchronize()


def time_c(before, reps=40):
    ts = []
    for _ in range(reps):
        if before is not None: run(before)
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record(); run(Cc); e1.record()
        ts.append((e0, e1))
    torch.cuda.synchronize()
    return float(np.median([a.elapsed_time(b) for a, b in ts])) * 1e3


for rnd in range(3):
    print(f"round {rnd}: C hot {time_c(None):.1f} us | after P (wrote X) {time_c(P):.1f} us | after Q (wrote elsewhere) {time_c(Q):.1f} us", flush=True)
