#!/usr/bin/env python3
"""Race screen of the halo2 tiles: one conv launched many times while a second stream keeps the GPU busy with unrelated GEMMs (and
without): every output must equal the first.  Prints mismatching launches per (layer, tile).
  python scripts/h2_stress.py [--tiles 55,61] [--reps 300] [--dtype f16]"""
import argparse, ctypes as C, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (os.path.join(ROOT, "yolo-v4-tf.keras_amd"), ROOT):
    sys.path.insert(0, p)
import torch
from yolo4hip import ext
ap = argparse.ArgumentParser()
ap.add_argument("--tiles", default="55,57,58,59,61,62"); ap.add_argument("--reps", type=int, default=300); ap.add_argument("--dtype", default="f16")
ap.add_argument("--batch", type=int, default=4); ap.add_argument("--act", type=int, default=1, help="1 leaky (the neck's convs), 2 mish")
a = ap.parse_args()
lib = ext.load()
dev, td, did = "cuda:0", {"bf16": torch.bfloat16, "f16": torch.float16}[a.dtype], ext.DTYPE_IDS[a.dtype]
LAYERS = [(19, 512, 1024), (38, 256, 512), (19, 512, 512)]
side = torch.cuda.Stream(device=dev)
junk = torch.randn(2048, 2048, device=dev)
total_bad = 0
for (s, cin, cout) in LAYERS:
    n = a.batch
    g = torch.Generator(device="cpu").manual_seed(s + cin)
    x = torch.randn((n, s, s, cin), generator=g).to(dev).to(td)
    w = (torch.randn((cout, cin, 3, 3), generator=g) * (1.0 / (3 * cin ** 0.5))).to(dev)
    cpad, nbytes = C.c_int32(), C.c_size_t()
    ext.check(lib.y4_packed_conv_bytes(did, cout, cin, 3, C.byref(cpad), C.byref(nbytes)))
    packed = torch.empty(nbytes.value, dtype=torch.uint8, device=dev); frag = torch.empty(nbytes.value, dtype=torch.uint8, device=dev)
    ext.check(lib.y4_pack_conv_weights(did, cout, cin, 3, ext.ptr(w), ext.ptr(packed), ext.stream_ptr()))
    ext.check(lib.y4_pack_conv_frag32(did, cout, cin, ext.ptr(packed), ext.ptr(frag), ext.stream_ptr()))
    sc = torch.ones(cpad.value, device=dev); sh = torch.zeros(cpad.value, device=dev)
    for tile in [int(t) for t in a.tiles.split(",")]:
        out = torch.zeros((n, s, s, cout), dtype=td, device=dev)
        d = ext.y4_conv_desc()
        d.dtype = did; d.n, d.h, d.w, d.cin = n, s, s, cin
        d.cout, d.ksize, d.stride, d.act = cout, 3, 1, a.act
        d.in_cstride, d.in_coff, d.out_cstride, d.out_coff = cin, 0, cout, 0
        d.in_ = x.data_ptr(); d.wt = packed.data_ptr(); d.scale = sc.data_ptr(); d.shift = sh.data_ptr(); d.out = out.data_ptr(); d.tile = tile
        d.wt_frag = frag.data_ptr()
        if lib.y4_conv2d(C.byref(d), ext.stream_ptr()) != 0:
            continue
        torch.cuda.synchronize()
        ref = out.clone()
        bad = {"quiet": 0, "loaded": 0}
        seen = set()
        for mode in ("quiet", "loaded"):
            for r in range(a.reps):
                out.zero_()
                if mode == "loaded":
                    with torch.cuda.stream(side):
                        for _ in range(2):
                            junk = (junk @ junk).clamp_(-1, 1)
                ext.check(lib.y4_conv2d(C.byref(d), ext.stream_ptr()))
                torch.cuda.current_stream().synchronize()
                if not torch.equal(out, ref):
                    bad[mode] += 1
                    seen.add(float(out.float().abs().sum().item()))
            side.synchronize()
        total_bad += bad["quiet"] + bad["loaded"]
        print(f"3x3 {cin}->{cout} @{s}^2 n{n} {a.dtype} tile {tile}: mismatching launches quiet {bad['quiet']}/{a.reps} loaded {bad['loaded']}/{a.reps}  distinct wrong outputs {len(seen)}", flush=True)
print("TOTAL mismatches", total_bad)
