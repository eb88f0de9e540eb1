#!/usr/bin/env python3
"""Layer benchmark of the halo2 tiles (conv_halo2_kernel.h: one wave per SIMD, v_mfma_32x32x16, weights in registers) against the halo
tiles and the implicit-GEMM tiles on the plan's 3x3 stride-1 layers at 608 x 608 / batch 32: us per launch in a hot loop (median of 5
blocks of 20 launches), TFLOP/s, equality with the 32x32x16 implicit-GEMM tile (same summation order) and the distance to a float32
torch conv of the same 16-bit operands.

  python scripts/halo2_bench.py [--dtype bf16] [--batch 32] [--cold 64] [--layers 0,1]
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

LAYERS = [  # (side, cin, cout, residual, other tiles to compare with)
    (38, 256, 512, False, (33, 19, 51)),
    (19, 512, 1024, False, (33, 19, 51)),
    (38, 256, 256, True, (33, 19, 51)),
    (19, 512, 512, True, (36, 20, 53)),
    (76, 128, 256, False, (33, 19, 54)),
    (76, 128, 128, True, (36, 38, 54)),
]


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--dtype", default="bf16")
    ap.add_argument("--batch", type=int, default=32)
    ap.add_argument("--reps", type=int, default=20)
    ap.add_argument("--cold", type=int, default=0)
    ap.add_argument("--layers", default="")
    ap.add_argument("--tiles", default="")
    ap.add_argument("--check", type=int, default=1)
    ap.add_argument("--fill", default="randn", help="randn | zero | small (operand values: power draw depends on the data's bit toggling)")
    a = ap.parse_args()
    import torch
    from yolo4hip import ext
    lib = ext.load()
    dev = "cuda:0"
    td = {"bf16": torch.bfloat16, "f16": torch.float16}[a.dtype]
    did = ext.DTYPE_IDS[a.dtype]
    ntiles = lib.y4_conv_tile_count()
    halo2 = []
    for t in range(1, ntiles + 1):
        cfg = (C.c_int32 * 6)()
        ext.check(lib.y4_conv_tile_desc(t, cfg))
        if cfg[5] == 21:
            halo2.append(t)
    if a.tiles:
        halo2 = [int(t) for t in a.tiles.split(",")]
    sel = [int(i) for i in a.layers.split(",")] if a.layers else range(len(LAYERS))
    for li in sel:
        side, cin, cout, use_res, cmp_tiles = LAYERS[li]
        n = a.batch
        g = torch.Generator(device="cpu").manual_seed(side + cin)
        x = torch.randn((n, side, side, cin), generator=g).to(dev).to(td)
        w = (torch.randn((cout, cin, 3, 3), generator=g) * (1.0 / (3 * cin ** 0.5))).to(dev)
        if a.fill == "zero":
            x.zero_(); w.zero_()
        elif a.fill == "small":
            x = (x.float().abs() * 0 + 1.0).to(td); w = w * 0 + 0.01
        res = torch.randn((n, side, side, cout), generator=g).to(dev).to(td) if use_res else None
        cpad, nbytes = C.c_int32(), C.c_size_t()
        ext.check(lib.y4_packed_conv_bytes(did, cout, cin, 3, C.byref(cpad), C.byref(nbytes)))
        packed = torch.empty(nbytes.value, dtype=torch.uint8, device=dev)
        frag = torch.empty(nbytes.value, dtype=torch.uint8, device=dev)
        ext.check(lib.y4_pack_conv_weights(did, cout, cin, 3, ext.ptr(w), ext.ptr(packed), ext.stream_ptr()))
        ext.check(lib.y4_pack_conv_frag32(did, cout, cin, ext.ptr(packed), ext.ptr(frag), ext.stream_ptr()))
        sc = (torch.rand(cpad.value, generator=g) + 0.5).to(dev)
        sh = (torch.randn(cpad.value, generator=g) * 0.1).to(dev)
        ref32 = None
        if a.check:
            # float32 reference of the same rounded operands: conv -> affine -> mish (+ residual)
            wq = w.to(td).float()
            y = torch.nn.functional.conv2d(x.float().permute(0, 3, 1, 2), wq, padding=1)
            y = y * sc[:cout].view(1, -1, 1, 1) + sh[:cout].view(1, -1, 1, 1)
            y = y * torch.tanh(torch.nn.functional.softplus(y))
            y = y.permute(0, 2, 3, 1)
            if res is not None:
                y = y + res.float()
            ref32 = y.contiguous()
        flops = 2.0 * n * side * side * cout * cin * 9
        results, ref = [], None
        for tile in list(cmp_tiles) + halo2:
            out = torch.zeros((n, side, side, cout), dtype=td, device=dev)
            d = ext.y4_conv_desc()
            d.dtype = did; d.n, d.h, d.w, d.cin = n, side, side, cin
            d.cout, d.ksize, d.stride, d.act = cout, 3, 1, 2
            d.in_cstride, d.in_coff, d.out_cstride, d.out_coff = cin, 0, cout, 0
            d.in_ = x.data_ptr(); d.wt = packed.data_ptr(); d.scale = sc.data_ptr(); d.shift = sh.data_ptr()
            d.out = out.data_ptr(); d.tile = tile
            d.wt_frag = frag.data_ptr() if tile in halo2 else None
            if res is not None:
                d.res = res.data_ptr(); d.res_cstride = cout; d.res_coff = 0
            rc = lib.y4_conv2d(C.byref(d), ext.stream_ptr())
            if rc != 0:
                print(f"  tile {tile}: {lib.y4_last_error().decode()}")
                continue
            torch.cuda.synchronize()
            if ref is None:
                ref = out.clone()
            same = bool(torch.equal(out, ref))
            err = float((out.float() - ref32).abs().max()) if ref32 is not None else -1.0
            times = []
            if a.cold:
                if "flush" not in globals():
                    globals()["flush"] = (torch.empty(a.cold << 20, dtype=torch.uint8, device=dev), torch.empty(a.cold << 20, dtype=torch.uint8, device=dev))
                for _ in range(15):
                    flush[1].copy_(flush[0])
                    xin = x.clone()
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
            # determinism: the timed launches wrote the same output
            lib.y4_conv2d(C.byref(d), ext.stream_ptr()); torch.cuda.synchronize()
            results.append((tile, us, flops / us / 1e6, same, err))
        tag = f"3x3 {cin}->{cout} @{side}^2{' +Add' if use_res else ''} b{n} {a.dtype}"
        print(tag + ": " + "  ".join(f"t{t}{'*' if t in halo2 else ''} {us:.1f}us {tf:.0f}TF{'' if same else ' DIFF'} err {err:.3g}" for t, us, tf, same, err in results), flush=True)


if __name__ == "__main__":
    main()
