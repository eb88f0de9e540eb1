#!/usr/bin/env python3
"""One 3x3 conv at batch 1 through y4_conv2d, hot loop and cold (weights + activations flushed from the L2s by a 512 MB copy between
launches): us per launch for a few latency tiles.  Run it in two trees to A/B a kernel change (scripts/b1_sched_ab.py is the
whole-network form)."""
import ctypes as C, os, sys
import numpy as np
ROOT = os.getcwd()
sys.path.insert(0, os.path.join(ROOT, "yolo-v4-tf.keras_amd"))
import torch
from yolo4hip import ext
lib = ext.load()
dev = "cuda:0"; td = torch.bfloat16; did = ext.DTYPE_IDS["bf16"]
big = torch.empty(512 << 20, dtype=torch.uint8, device=dev); big2 = torch.empty_like(big)
for side, cin, cout, tiles in ((19, 512, 512, (49, 149, 48, 50, 20)), (38, 256, 512, (49, 149, 50, 43)), (19, 512, 1024, (49, 149, 50))):
    g = torch.Generator().manual_seed(1)
    x = torch.randn((1, side, side, cin), generator=g).to(dev).to(td)
    w = (torch.randn((cout, cin, 3, 3), generator=g) * 0.02).to(dev)
    cpad, nbytes = C.c_int32(), C.c_size_t()
    ext.check(lib.y4_packed_conv_bytes(did, cout, cin, 3, C.byref(cpad), C.byref(nbytes)))
    packed = torch.empty(nbytes.value, dtype=torch.uint8, device=dev)
    ext.check(lib.y4_pack_conv_weights(did, cout, cin, 3, ext.ptr(w), ext.ptr(packed), ext.stream_ptr()))
    sc = torch.ones(cpad.value, dtype=torch.float32, device=dev); sh = torch.zeros(cpad.value, dtype=torch.float32, device=dev)
    ws = torch.zeros(16 * 1024 + 32 * 1024 * 1024, dtype=torch.uint8, device=dev)
    line = f"3x3 {cin}->{cout} @{side}^2 b1:"
    for tile in tiles:
        out = torch.zeros((1, side, side, cout), dtype=td, device=dev)
        d = ext.y4_conv_desc()
        d.dtype = did; d.n, d.h, d.w, d.cin = 1, side, side, cin
        d.cout, d.ksize, d.stride, d.act = cout, 3, 1, 2
        d.in_cstride, d.in_coff, d.out_cstride, d.out_coff = cin, 0, cout, 0
        d.in_ = x.data_ptr(); d.wt = packed.data_ptr(); d.scale = sc.data_ptr(); d.shift = sh.data_ptr(); d.out = out.data_ptr(); d.tile = tile
        d.splitk_ws = ws.data_ptr(); d.splitk_ws_bytes = ws.numel()
        if lib.y4_conv2d(C.byref(d), ext.stream_ptr()) != 0: continue
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(200): lib.y4_conv2d(C.byref(d), ext.stream_ptr())
        e1.record(); e1.synchronize(); hot = e0.elapsed_time(e1) / 200 * 1e3
        cold = []
        for _ in range(12):
            big2.copy_(big)
            e0.record(); lib.y4_conv2d(C.byref(d), ext.stream_ptr()); e1.record(); e1.synchronize()
            cold.append(e0.elapsed_time(e1) * 1e3)
        line += f"  t{tile} hot {hot:.1f} cold {np.median(cold):.1f}"
    print(line)
