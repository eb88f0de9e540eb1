#!/usr/bin/env python3
"""Runs ONE conv tile on ONE of halo2_bench.py's layers a few times (what a rocprofv3 --pmc pass wraps).  env: H2_LAYER, H2_TILE, H2_REPS"""
import ctypes as C, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (os.path.join(ROOT, "yolo-v4-tf.keras_amd"), ROOT, os.path.join(ROOT, "scripts")):
    sys.path.insert(0, p)
import torch
from yolo4hip import ext
from halo2_bench import LAYERS
lib = ext.load()
li, tile, reps = int(os.environ.get("H2_LAYER", 0)), int(os.environ.get("H2_TILE", 55)), int(os.environ.get("H2_REPS", 5))
dt = os.environ.get("H2_DTYPE", "bf16")
side, cin, cout, use_res, _ = LAYERS[li]
n, dev, td, did = 32, "cuda:0", {"bf16": torch.bfloat16, "f16": torch.float16}[dt], ext.DTYPE_IDS[dt]
g = torch.Generator(device="cpu").manual_seed(side + cin)
x = torch.randn((n, side, side, cin), generator=g).to(dev).to(td)
w = (torch.randn((cout, cin, 3, 3), generator=g) * (1.0 / (3 * cin ** 0.5))).to(dev)
res = torch.randn((n, side, side, cout), generator=g).to(dev).to(td) if use_res else None
cpad, nbytes = C.c_int32(), C.c_size_t()
ext.check(lib.y4_packed_conv_bytes(did, cout, cin, 3, C.byref(cpad), C.byref(nbytes)))
packed = torch.empty(nbytes.value, dtype=torch.uint8, device=dev); frag = torch.empty(nbytes.value, dtype=torch.uint8, device=dev)
ext.check(lib.y4_pack_conv_weights(did, cout, cin, 3, ext.ptr(w), ext.ptr(packed), ext.stream_ptr()))
ext.check(lib.y4_pack_conv_frag32(did, cout, cin, ext.ptr(packed), ext.ptr(frag), ext.stream_ptr()))
sc = torch.ones(cpad.value, device=dev); sh = torch.zeros(cpad.value, device=dev)
out = torch.zeros((n, side, side, cout), dtype=td, device=dev)
d = ext.y4_conv_desc()
d.dtype = did; d.n, d.h, d.w, d.cin = n, side, side, cin
d.cout, d.ksize, d.stride, d.act = cout, 3, 1, 2
d.in_cstride, d.in_coff, d.out_cstride, d.out_coff = cin, 0, cout, 0
d.in_ = x.data_ptr(); d.wt = packed.data_ptr(); d.scale = sc.data_ptr(); d.shift = sh.data_ptr(); d.out = out.data_ptr(); d.tile = tile
d.wt_frag = frag.data_ptr()
if res is not None:
    d.res = res.data_ptr(); d.res_cstride = cout; d.res_coff = 0
for _ in range(reps):
    ext.check(lib.y4_conv2d(C.byref(d), ext.stream_ptr()))
torch.cuda.synchronize()
