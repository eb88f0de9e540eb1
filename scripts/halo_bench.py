#!/usr/bin/env python3
"""Layer benchmark of the halo tiles (conv_halo_kernel.h) against the implicit-GEMM tiles on the plan's 3x3 stride-1 layers at
608 x 608 / batch 32: us per launch in a hot loop (median of 5 blocks of 20 launches), TFLOP/s, and bit-identity of the outputs.

  python scripts/halo_bench.py [--dtype bf16] [--batch 32]
"""
import argparse
import ctypes as C
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (os.path.join(ROOT, "yolo-v4-tf.keras_amd"), ROOT, os.path.join(ROOT, "tests")):
    if p not in sys.path:
        sys.path.insert(0, p)

LAYERS = [  # (side, cin, cout, residual, igemm tiles to compare with)
    (38, 256, 512, False, (19, 39, 41, 38)),
    (19, 512, 1024, False, (19, 39, 20)),
    (38, 256, 256, True, (19, 38, 20)),
    (19, 512, 512, True, (20, 19)),
    (76, 128, 256, False, (19, 38, 18)),
    (76, 128, 128, True, (8, 38, 20)),
]


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--dtype", default="bf16")
    ap.add_argument("--batch", type=int, default=32)
    ap.add_argument("--reps", type=int, default=20)
    ap.add_argument("--cold", type=int, default=0, help="MB copied between single timed launches (evicts the L2s; > 256 also the Infinity Cache): what a launch costs IN a step rather than in a hot loop")
    a = ap.parse_args()
    import torch
    from yolo4hip import ext
    lib = ext.load()
    dev = "cuda:0"
    td = {"bf16": torch.bfloat16, "f16": torch.float16}[a.dtype]
    did = ext.DTYPE_IDS[a.dtype]
    ntiles = lib.y4_conv_tile_count()
    halo = []
    for t in range(1, ntiles + 1):
        cfg = (C.c_int32 * 6)()
        ext.check(lib.y4_conv_tile_desc(t, cfg))
        if cfg[5] == 20:
            halo.append(t)
    for side, cin, cout, use_res, cmp_tiles in LAYERS:
        n = a.batch
        g = torch.Generator(device="cpu").manual_seed(side + cin)
        x = torch.randn((n, side, side, cin), generator=g).to(dev).to(td)
        w = (torch.randn((cout, cin, 3, 3), generator=g) * (1.0 / (3 * cin ** 0.5))).to(dev)
        res = torch.randn((n, side, side, cout), generator=g).to(dev).to(td) if use_res else None
        cpad, nbytes = C.c_int32(), C.c_size_t()
        ext.check(lib.y4_packed_conv_bytes(did, cout, cin, 3, C.byref(cpad), C.byref(nbytes)))
        packed = torch.empty(nbytes.value, dtype=torch.uint8, device=dev)
        ext.check(lib.y4_pack_conv_weights(did, cout, cin, 3, ext.ptr(w), ext.ptr(packed), ext.stream_ptr()))
        sc = torch.ones(cpad.value, dtype=torch.float32, device=dev)
        sh = torch.zeros(cpad.value, dtype=torch.float32, device=dev)
        flops = 2.0 * n * side * side * cout * cin * 9
        results, ref = [], None
        for tile in list(cmp_tiles) + halo:
            out = torch.zeros((n, side, side, cout), dtype=td, device=dev)
            d = ext.y4_conv_desc()
            d.dtype = did; d.n, d.h, d.w, d.cin = n, side, side, cin
            d.cout, d.ksize, d.stride, d.act = cout, 3, 1, 2
            d.in_cstride, d.in_coff, d.out_cstride, d.out_coff = cin, 0, cout, 0
            d.in_ = x.data_ptr(); d.wt = packed.data_ptr(); d.scale = sc.data_ptr(); d.shift = sh.data_ptr()
            d.out = out.data_ptr(); d.tile = tile
            if res is not None:
                d.res = res.data_ptr(); d.res_cstride = cout; d.res_coff = 0
            rc = lib.y4_conv2d(C.byref(d), ext.stream_ptr())
            if rc != 0:
                continue
            torch.cuda.synchronize()
            if ref is None:
                ref = out.clone()
            same = bool(torch.equal(out, ref))
            times = []
            if a.cold:
                if "flush" not in globals():
                    globals()["flush"] = (torch.empty(a.cold << 20, dtype=torch.uint8, device=dev), torch.empty(a.cold << 20, dtype=torch.uint8, device=dev))
                for _ in range(15):
                    flush[1].copy_(flush[0])
                    xin = x.clone()                      # the input as the previous layer leaves it: freshly written
                    d.in_ = xin.data_ptr()
                    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                    e0.record(); lib.y4_conv2d(C.byref(d), ext.stream_ptr()); e1.record(); e1.synchronize()
                    times.append(e0.elapsed_time(e1) * 1e3)
                d.in_ = x.data_ptr()
            else:
              for _ in range(5):
                e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                e0.record()
                for _ in range(a.reps):
                    lib.y4_conv2d(C.byref(d), ext.stream_ptr())
                e1.record()
                e1.synchronize()
                times.append(e0.elapsed_time(e1) / a.reps * 1e3)
            us = float(np.median(times))
            results.append((tile, us, flops / us / 1e6, same))
        tag = f"3x3 {cin}->{cout} @{side}^2{' +Add' if use_res else ''} b{n} {a.dtype}"
        print(tag + ": " + "  ".join(f"t{t}{'h' if t in halo else ''} {us:.1f}us {tf:.0f}TF{'' if same else ' DIFF'}" for t, us, tf, same in results))


if __name__ == "__main__":
    main()
