"""Micro-benchmark of y4_conv2d over (shape x tile): device time via events on the launch stream."""
import sys, os, ctypes as C
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "yolo-v4-tf.keras_amd")); sys.path.insert(0, ROOT)
import numpy as np, torch
from yolo4hip import ext
lib = ext.load()
dev = "cuda:0"
dtype = sys.argv[1] if len(sys.argv) > 1 else "bf16"
did = ext.DTYPE_IDS[dtype]; td = {"bf16": torch.bfloat16, "f16": torch.float16, "f32": torch.float32}[dtype]
N = int(os.environ.get("NB", "32"))
# k, s, cin, cout, side(in), res
SHAPES = [(1,1,64,64,304,0),(1,1,128,64,304,0),(3,2,32,64,608,0),(3,1,32,64,304,1),(1,1,128,128,76,0),(3,1,128,128,76,1),(1,1,256,256,38,0),(3,1,256,256,38,1),
          (1,1,512,512,19,0),(3,1,512,512,19,1),(3,1,512,1024,19,0),(1,1,1024,512,19,0),(3,1,256,512,38,0),(1,1,512,256,38,0),(3,1,128,256,76,0),(1,1,256,128,76,0),(1,1,256,255,76,0)]
if len(sys.argv) > 2: SHAPES = [tuple(int(v) for v in sys.argv[2].split(","))]
ntiles = lib.y4_conv_tile_count()
for (k, s, cin, cout, side, res) in SHAPES:
    x = torch.randn((N, side, side, cin), device=dev).to(td)
    so = side // s
    out = torch.empty((N, so, so, (cout + 7)//8*8), device=dev, dtype=td)
    r = torch.randn((N, so, so, cout), device=dev).to(td) if res else None
    cpad, nb = C.c_int32(), C.c_size_t()
    ext.check(lib.y4_packed_conv_bytes(did, cout, cin, k, C.byref(cpad), C.byref(nb)))
    w = torch.randn((cout, cin, k, k), device=dev) * 0.05
    packed = torch.empty(nb.value, dtype=torch.uint8, device=dev)
    ext.check(lib.y4_pack_conv_weights(did, cout, cin, k, ext.ptr(w), ext.ptr(packed), ext.stream_ptr()))
    sc = torch.ones(cpad.value, device=dev); sh = torch.zeros(cpad.value, device=dev)
    d = ext.y4_conv_desc(); d.dtype = did; d.n, d.h, d.w, d.cin = N, side, side, cin
    d.cout, d.ksize, d.stride, d.act = cout, k, s, 2
    d.in_cstride, d.out_cstride = cin, (cout + 7)//8*8
    d.in_ = x.data_ptr(); d.wt = packed.data_ptr(); d.scale = sc.data_ptr(); d.shift = sh.data_ptr(); d.out = out.data_ptr()
    if res: d.res = r.data_ptr(); d.res_cstride = cout
    flops = 2.0 * k * k * cin * cout * so * so * N
    byts = (x.numel() + out.numel() + (r.numel() if res else 0)) * x.element_size()
    line = f"k{k}s{s} {cin:4d}->{cout:4d} @{side:3d} {'+res' if res else '    '} {flops/1e9:7.1f}GF {byts/1e6:7.1f}MB |"
    for tile in range(0, ntiles + 1):
        d.tile = tile
        if lib.y4_conv2d(C.byref(d), ext.stream_ptr()) != 0:
            line += f" t{tile}:  --  "; continue
        for _ in range(3): lib.y4_conv2d(C.byref(d), ext.stream_ptr())
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(10): lib.y4_conv2d(C.byref(d), ext.stream_ptr())
        e1.record(); torch.cuda.synchronize()
        ms = e0.elapsed_time(e1) / 10
        line += f" t{tile}:{ms*1e3:6.1f}us"
    best = min(float(t.split(':')[1][:-2]) for t in line.split('|')[1].split(' t')[1:] if '--' not in t)
    print(line, f"| best {flops/best/1e6:6.0f} TF {byts/best/1e6:5.2f} TB/s", flush=True)
