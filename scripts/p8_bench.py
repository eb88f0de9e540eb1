"""Phased conv kernel (conv_p8_kernel.h) against the plain tiles, layer shapes of the 608^2 batch-32 step, random data:
interleaved rounds in one process, median device time per launch (HIP events on the launch stream).
usage: p8_bench.py [tile ids ...]   (default: 19 39 41 42 = plain, staggered, software-pipelined, producer/consumer)"""
import sys, os, ctypes as C
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "yolo-v4-tf.keras_amd")); sys.path.insert(0, ROOT)
import numpy as np, torch
from yolo4hip import ext
lib = ext.load()
dev, td, did = "cuda:0", torch.bfloat16, ext.DTYPE_IDS["bf16"]
N = int(os.environ.get("NB", "32"))
TILES = [int(a) for a in sys.argv[1:]] or [19, 39, 41, 42]
# k, s, cin, cout, side(in), res
SHAPES = [(3,1,256,512,38,0),(1,1,512,256,38,0),(3,1,256,256,38,1),(1,1,256,256,38,0),(3,1,512,1024,19,0),(1,1,1024,512,19,0),(3,1,512,512,19,1),
          (1,1,2048,512,19,0),(3,1,128,256,76,0),(1,1,256,256,76,0),(3,2,256,512,76,0),(3,2,512,1024,38,0)]
ROUNDS, PER = 5, 10
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
    times = {t: [] for t in TILES}
    ok = {}
    for t in TILES:
        d.tile = t
        ok[t] = lib.y4_conv2d(C.byref(d), ext.stream_ptr()) == 0
    torch.cuda.synchronize()
    for _ in range(ROUNDS):
        for t in TILES:
            if not ok[t]: continue
            d.tile = t
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            lib.y4_conv2d(C.byref(d), ext.stream_ptr())
            e0.record()
            for _ in range(PER): lib.y4_conv2d(C.byref(d), ext.stream_ptr())
            e1.record(); torch.cuda.synchronize()
            times[t].append(e0.elapsed_time(e1) / PER * 1e3)
    line = f"k{k}s{s} {cin:4d}->{cout:4d} @{side:3d}{'+res' if res else '    '} {flops/1e9:6.1f}GF |"
    for t in TILES:
        if not ok[t]: line += f" t{t}:   --      "; continue
        med = float(np.median(times[t]))
        line += f" t{t}:{med:6.1f}us {flops/med/1e6:5.0f}TF"
    print(line, flush=True)
