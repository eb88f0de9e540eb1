"""Where the hand-written conv kernels stand against the vendor libraries on the same layer shapes (608^2, batch 32, bf16,
random data): per shape the best of all tuner-eligible tiles of y4_conv2d (BN scale/shift + Mish epilogue included) against
  * torch.nn.functional.conv2d on channels_last bf16 tensors (MIOpen; bare convolution, no epilogue), and
  * for the 1x1 layers torch.matmul [M,K]x[K,N] (hipBLASLt; bare GEMM), for the 3x3 layers the GEMM of the same M, N and
    K = 9*Cin (what a library GEMM reaches if im2col were free -- an upper bound, not an implementation).
Median device time of 5 rounds x 10 launches, interleaved in one process, HIP events on the launch stream.
A diagnostic: nothing in the product path calls these libraries."""
import sys, os, ctypes as C
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "yolo-v4-tf.keras_amd")); sys.path.insert(0, ROOT)
import numpy as np, torch
import torch.nn.functional as F
from yolo4hip import ext
lib = ext.load()
torch.backends.cudnn.benchmark = True                    # MIOpen: search for its best solver per shape
dev, td, did = "cuda:0", torch.bfloat16, ext.DTYPE_IDS["bf16"]
N = int(os.environ.get("NB", "32"))
NT = lib.y4_conv_tile_count()
# k, s, cin, cout, side(in), res
SHAPES = [(3,1,64,64,152,1),(1,1,128,128,152,0),(3,2,64,128,304,0),(3,1,128,128,76,1),(1,1,128,128,76,0),(3,2,128,256,152,0),
          (3,1,256,256,38,1),(1,1,256,256,38,0),(3,2,256,512,76,0),(3,1,512,512,19,1),(1,1,512,512,19,0),(3,2,512,1024,38,0),
          (3,1,512,1024,19,0),(1,1,1024,512,19,0),(1,1,2048,512,19,0),(3,1,256,512,38,0),(1,1,512,256,38,0),(3,1,128,256,76,0),(1,1,256,128,76,0)]
ROUNDS, PER = 5, 10


def timed(fn):
    fn(); torch.cuda.synchronize()
    ts = []
    for _ in range(ROUNDS):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        fn(); e0.record()
        for _ in range(PER): fn()
        e1.record(); torch.cuda.synchronize()
        ts.append(e0.elapsed_time(e1) / PER * 1e3)
    return float(np.median(ts))


print(f"{'layer':32s} {'GF':>6s} | {'y4 best tile':>22s} | {'MIOpen conv2d':>16s} | {'hipBLASLt GEMM':>16s}")
for (k, s, cin, cout, side, res) in SHAPES:
    x = torch.randn((N, side, side, cin), device=dev).to(td)
    so = side // s
    out = torch.empty((N, so, so, cout), device=dev, dtype=td)
    r = torch.randn((N, so, so, cout), device=dev).to(td) if res else None
    cpad, nb = C.c_int32(), C.c_size_t()
    ext.check(lib.y4_packed_conv_bytes(did, cout, cin, k, C.byref(cpad), C.byref(nb)))
    w = torch.randn((cout, cin, k, k), device=dev) * 0.05
    packed = torch.empty(nb.value, dtype=torch.uint8, device=dev)
    ext.check(lib.y4_pack_conv_weights(did, cout, cin, k, ext.ptr(w), ext.ptr(packed), ext.stream_ptr()))
    sc = torch.ones(cpad.value, device=dev); sh = torch.zeros(cpad.value, device=dev)
    d = ext.y4_conv_desc(); d.dtype = did; d.n, d.h, d.w, d.cin = N, side, side, cin
    d.cout, d.ksize, d.stride, d.act = cout, k, s, 1
    d.in_cstride, d.out_cstride = cin, cout
    d.in_ = x.data_ptr(); d.wt = packed.data_ptr(); d.scale = sc.data_ptr(); d.shift = sh.data_ptr(); d.out = out.data_ptr()
    if res: d.res = r.data_ptr(); d.res_cstride = cout
    flops = 2.0 * k * k * cin * cout * so * so * N
    best, best_t = 1e30, 0
    watch = {int(a): None for a in os.environ.get("WATCH", "").split(",") if a}
    for t in range(1, NT + 1):
        tc = ext.tile_cfg(t) if hasattr(ext, "tile_cfg") else None
        d.tile = t
        if lib.y4_conv2d(C.byref(d), ext.stream_ptr()) != 0: continue
        if t in (33, 34, 35, 36, 37, 42): continue                       # not offered by the tuner
        us = timed(lambda: lib.y4_conv2d(C.byref(d), ext.stream_ptr()))
        if us < best: best, best_t = us, t
        if t in watch: watch[t] = us
    # MIOpen: NCHW-shaped tensors in channels_last memory format = the same NHWC bytes
    xc = x.permute(0, 3, 1, 2)                                          # [N,C,H,W] view over NHWC storage
    wc = w.to(td).contiguous(memory_format=torch.channels_last)
    pad = 1 if k == 3 and s == 1 else 0
    if k == 3 and s == 2:
        xin = F.pad(xc, (1, 0, 1, 0)).contiguous(memory_format=torch.channels_last)   # the reference's ZeroPadding2D(((1,0),(1,0)))
        conv = lambda: F.conv2d(xin, wc, stride=2)
    else:
        conv = lambda: F.conv2d(xc, wc, stride=1, padding=pad)
    try:
        us_mi = timed(conv)
    except Exception as e:                                              # noqa
        us_mi = float("nan")
    M, K = N * so * so, k * k * cin
    a = torch.randn((M, K), device=dev).to(td); b = torch.randn((K, cout), device=dev).to(td)
    us_mm = timed(lambda: torch.matmul(a, b))
    tf = lambda us: flops / us / 1e6
    print(f"k{k}s{s} {cin:4d}->{cout:4d} @{side:3d}{'+res' if res else '    '} M={M:6d} {flops/1e9:6.1f} | t{best_t:2d} {best:7.1f}us {tf(best):5.0f}TF | "
          f"{us_mi:7.1f}us {tf(us_mi):5.0f}TF | {us_mm:7.1f}us {tf(us_mm):5.0f}TF{'' if k == 1 else ' (bound)'}", flush=True)
    if watch:
        print("      " + "  ".join(f"t{t}: {'--' if u is None else f'{u:.1f}us'}" for t, u in watch.items()), flush=True)
    del a, b, x, out
