"""GPU parity of the implicit-GEMM conv kernel (y4_conv2d through the C ABI) against the oracle's
conv() unit (oracle/forward.py conv_block <- reference custom_layers.py:5-31,44).

Tolerances (stated per dtype):
  f32  : |gpu - oracle| <= 2e-5 + 2e-5*|oracle|  (both are fp32 fmaf/sum chains in different orders)
  bf16 : inputs/weights/residual are first rounded to bf16 and the oracle computes on those in fp32;
         the GPU result is rounded to bf16 on store -> 1 bf16 ulp (2^-8 relative) + 1e-2 abs slack
  f16  : same with fp16 (2^-10 relative) + 2e-3 abs
"""
import numpy as np
import pytest

from helpers import make_conv_weights, quantize, run_conv_gpu

pytestmark = pytest.mark.gpu

TOL = {"f32": (2e-5, 2e-5), "bf16": (1e-2, 2.0 ** -7), "f16": (2e-3, 2.0 ** -9)}

# k, stride, cin, cout, side, act, bn, residual, upsample, out_f32, in_pad, out_pad, n
CASES = [
    (1, 1, 64, 64, 24, "mish", True, False, False, False, (0, 0), (0, 0), 2),
    (1, 1, 64, 32, 20, "mish", True, False, False, False, (0, 0), (0, 0), 1),
    (3, 1, 32, 64, 20, "mish", True, True, False, False, (0, 0), (0, 0), 2),
    (3, 2, 32, 64, 32, "leaky", True, False, False, False, (0, 0), (0, 0), 1),
    (3, 1, 128, 256, 19, "leaky", True, False, False, False, (0, 0), (0, 0), 2),
    (3, 2, 256, 512, 38, "mish", True, False, False, False, (0, 0), (0, 0), 1),
    (1, 1, 512, 255, 19, None, False, False, False, True, (0, 0), (0, 0), 2),
    (1, 1, 256, 24, 13, None, False, False, False, True, (0, 0), (0, 0), 3),
    (1, 1, 512, 256, 13, "leaky", True, False, True, False, (0, 0), (256, 0), 2),
    (1, 1, 2048, 512, 13, "leaky", True, False, False, False, (0, 0), (0, 0), 1),
    (1, 1, 128, 64, 26, "mish", True, False, False, False, (64, 64), (64, 0), 1),
    (3, 1, 64, 64, 26, "mish", True, True, False, False, (0, 64), (0, 64), 1),
    (3, 2, 128, 256, 26, "leaky", True, False, False, False, (0, 0), (0, 256), 2),
]


def _ref(x, cw, k, stride, act, residual, upsample):
    from oracle.forward import conv_block
    y = conv_block(x, cw, k, stride, act, residual)
    if upsample:
        y = y.repeat(2, axis=1).repeat(2, axis=2)       # UpSampling2D nearest x2 (custom_layers.py:147)
    return y


@pytest.mark.parametrize("dtype", ["f32", "bf16", "f16"])
@pytest.mark.parametrize("case", CASES, ids=lambda c: f"k{c[0]}s{c[1]}_{c[2]}to{c[3]}_h{c[4]}")
def test_conv_vs_oracle(case, dtype):
    from yolo4hip.weights import ConvWeights
    k, stride, cin, cout, side, act, bn, use_res, ups, out_f32, in_pad, out_pad, n = case
    rng = np.random.default_rng(hash((k, stride, cin, cout, side)) % (2 ** 31))
    x = quantize(rng.standard_normal((n, side, side, cin)).astype(np.float32), dtype)
    cw = make_conv_weights(rng, cout, cin, k, bn)
    cwq = ConvWeights(w=quantize(cw.w, dtype), bn=cw.bn, bias=cw.bias)
    so = side // stride
    res = quantize(rng.standard_normal((n, so, so, cout)).astype(np.float32), dtype) if use_res else None
    got, full = run_conv_gpu(x, cwq, k, stride, act, dtype, residual=res, upsample=ups, out_f32=out_f32,
                             in_pad=in_pad, out_pad=out_pad)
    want = _ref(x, cwq, k, stride, act, res, ups)
    atol, rtol = TOL["f32"] if (out_f32 and dtype == "f32") else TOL[dtype]
    if out_f32 and dtype != "f32":
        atol, rtol = 1e-4, 1e-4      # fp32 store of an fp32 accumulator over 16-bit inputs
    err = np.abs(got - want)
    assert np.all(err <= atol + rtol * np.abs(want)), f"max err {err.max():.3e} at {np.unravel_index(err.argmax(), err.shape)}"
    # the kernel must not touch channels outside its slice (pad channels within the 8-rounded store excepted)
    cstore = (cout + 7) // 8 * 8
    if out_pad[0]:
        assert np.all(full[..., :out_pad[0]] == -5.0)
    if out_pad[1]:
        assert np.all(full[..., out_pad[0] + cstore:] == -5.0)


@pytest.mark.parametrize("dtype", ["f32", "bf16"])
def test_all_tiles_agree(dtype):
    """Every tile configuration that fits the shape gives the same result (same K order -> bitwise), per MFMA shape."""
    from yolo4hip import ext
    from yolo4hip.weights import ConvWeights
    rng = np.random.default_rng(5)
    x = quantize(rng.standard_normal((2, 17, 17, 128)).astype(np.float32), dtype)
    cw = make_conv_weights(rng, 128, 128, 3)
    cwq = ConvWeights(w=quantize(cw.w, dtype), bn=cw.bn)
    base, _ = run_conv_gpu(x, cwq, 3, 1, "mish", dtype)
    import ctypes as C
    lib = ext.load()
    ntiles = lib.y4_conv_tile_count()
    ran, base32, ran32 = 0, None, 0
    atol, rtol = TOL[dtype]
    for tile in range(1, ntiles + 1):
        cfg = (C.c_int32 * 6)()
        ext.check(lib.y4_conv_tile_desc(tile, cfg))
        try:
            got, _ = run_conv_gpu(x, cwq, 3, 1, "mish", dtype, tile=tile)
        except ext.Y4Error as e:
            assert e.code == -22      # tile does not fit this cin/cout / dtype: refused loudly, not computed wrongly
            continue
        if cfg[5] == 21 and cfg[4] == 64:      # halo2 tiles that walk 32-channel half chunks: a third order, held to the tolerance only
            err = np.abs(got - base)
            assert np.all(err <= 2 * atol + 2 * rtol * np.abs(base)), f"halo2 tile {tile}: {err.max():.3e} off the 16x16 tiles"
            continue
        if cfg[5] == 32 or cfg[5] == 21:       # 32x32x16 MFMA (implicit GEMM, and the halo2 tiles over whole chunks): another fp32 summation order; these agree among themselves
            if base32 is None: base32 = got
            assert np.array_equal(got, base32), f"32x32 tile {tile} differs from the first one: {np.abs(got - base32).max()}"
            err = np.abs(got - base)
            assert np.all(err <= 2 * atol + 2 * rtol * np.abs(base)), f"32x32 tile {tile}: {err.max():.3e} off the 16x16 tiles"
            ran32 += 1
            continue
        ran += 1
        assert np.array_equal(got, base), f"tile {tile} differs: {np.abs(got - base).max()}"
    assert ran >= 4 and (dtype == "f32" or ran32 >= 2)


HALO_CASES = [   # (side, cin, cout, n, act, residual, in_pad, out_pad): 3x3 stride-1 convs of the plan's 19^2 / 38^2 / 76^2 stages and smaller
    (19, 128, 256, 3, "leaky", False, (0, 0), (0, 0)),        # one band = the whole image
    (38, 64, 128, 2, "mish", True, (0, 0), (0, 0)),           # ragged bands (10 + 10 + 10 + 8 rows at 384 pixels), residual Add
    (38, 256, 256, 1, "mish", True, (64, 0), (0, 128)),       # four chunks: the halo double buffer turns over; channel slices
    (76, 128, 128, 1, "leaky", False, (0, 0), (0, 0)),        # pitch 80; 320-pixel bands of 4 rows
    (13, 128, 128, 5, "mish", False, (0, 0), (128, 0)),       # 416 / 32: pitch 16
    (26, 192, 256, 2, "leaky", False, (0, 64), (0, 0)),       # three chunks
    (24, 64, 128, 2, "mish", True, (0, 0), (0, 0)),           # one chunk (Cin = 64); two bands of 12 rows
]


@pytest.mark.parametrize("dtype", ["bf16", "f16"])
@pytest.mark.parametrize("case", HALO_CASES, ids=lambda c: f"h{c[0]}_{c[1]}to{c[2]}_n{c[3]}")
def test_halo_tiles_vs_oracle_and_bit_identical(case, dtype):
    """conv_halo_kernel (tile ids with schedule code 20: the input halo tile staged once per 64-channel chunk, nine taps from LDS)
    against the oracle's conv_block (reference custom_layers.py:5-31, :44) AND bit for bit against the implicit-GEMM kernel's
    built-in tile: same MFMAs in the same canonical K order (csrc/common.h)."""
    import ctypes as C
    from yolo4hip import ext
    from yolo4hip.weights import ConvWeights
    side, cin, cout, n, act, use_res, in_pad, out_pad = case
    rng = np.random.default_rng(1000 + side + cin)
    x = quantize(rng.standard_normal((n, side, side, cin)).astype(np.float32), dtype)
    cw = make_conv_weights(rng, cout, cin, 3)
    cwq = ConvWeights(w=quantize(cw.w, dtype), bn=cw.bn, bias=cw.bias)
    res = quantize(rng.standard_normal((n, side, side, cout)).astype(np.float32), dtype) if use_res else None
    base, _ = run_conv_gpu(x, cwq, 3, 1, act, dtype, residual=res, in_pad=in_pad, out_pad=out_pad)
    want = _ref(x, cwq, 3, 1, act, res, False)
    atol, rtol = TOL[dtype]
    assert np.all(np.abs(base - want) <= atol + rtol * np.abs(want))
    lib = ext.load()
    ran = 0
    for tile in range(1, lib.y4_conv_tile_count() + 1):
        cfg = (C.c_int32 * 6)()
        ext.check(lib.y4_conv_tile_desc(tile, cfg))
        if cfg[5] != 20:
            continue
        try:
            got, full = run_conv_gpu(x, cwq, 3, 1, act, dtype, residual=res, in_pad=in_pad, out_pad=out_pad, tile=tile)
        except ext.Y4Error as e:
            assert e.code == -22          # this band geometry does not fit the layer: refused, not computed wrongly
            continue
        ran += 1
        err = np.abs(got - want)
        assert np.all(err <= atol + rtol * np.abs(want)), f"halo tile {tile}: max err {err.max():.3e} at {np.unravel_index(err.argmax(), err.shape)}"
        assert np.array_equal(got, base), f"halo tile {tile} differs from the implicit-GEMM kernel: {np.abs(got - base).max():.3e}"
        if out_pad[0]:
            assert np.all(full[..., :out_pad[0]] == -5.0)
        if out_pad[1]:
            assert np.all(full[..., out_pad[0] + cout:] == -5.0)
    assert ran >= 1, "no halo tile fits this case"


HALO2_CASES = HALO_CASES + [
    (38, 256, 512, 2, "mish", False, (0, 0), (0, 0)),         # the plan's 3x3 256 -> 512 @38^2: four chunks, four channel tiles
    (19, 512, 512, 3, "mish", True, (0, 0), (0, 0)),          # eight chunks, residual Add, 19 x 21 / 19 x 23 halo rows
    (24, 128, 128, 3, "linear", False, (0, 0), (0, 0)),       # a width that is no stage of the plan; the general (linear) epilogue
    (52, 128, 256, 1, "leaky", False, (64, 64), (128, 0)),    # 416 / 8; channel slices on both sides
    (38, 128, 200, 5, "mish", False, (0, 0), (0, 56)),        # Cout = 200: the second channel tile is partly padding rows (general epilogue,
                                                              # channel predicates); 5 images x 4 bands x 2 tiles = 40 workgroups: not a multiple of 8 XCDs
    (19, 64, 128, 7, "leaky", True, (0, 0), (0, 0)),          # one chunk of K only (Cin = 64): prologue -> one pass of the body -> drain; 7 workgroups
]


@pytest.mark.parametrize("dtype", ["bf16", "f16"])
@pytest.mark.parametrize("case", HALO2_CASES, ids=lambda c: f"h{c[0]}_{c[1]}to{c[2]}_n{c[3]}")
def test_halo2_tiles_vs_oracle(case, dtype):
    """conv_halo2_kernel (tile ids with schedule code 21: one wave per SIMD, v_mfma_f32_32x32x16, halo tile in LDS, weights read into
    registers in MFMA-fragment order) against the oracle's conv_block (reference custom_layers.py:5-31, :44) at the tolerance of every
    other tile.  These tiles sum the K axis in k-steps of 16: the tiles that walk whole 64-channel chunks do so in the order of the
    implicit-GEMM kernel's 32x32x16 tiles and equal those bit for bit; the ones that walk half chunks (128-byte K rows in the tile
    table = KC 64, 64-byte = KC 32) are their own order.  Every tile must give the same bits twice."""
    import ctypes as C
    from yolo4hip import ext
    from yolo4hip.weights import ConvWeights
    side, cin, cout, n, act, use_res, in_pad, out_pad = case
    rng = np.random.default_rng(2000 + side + cin)
    x = quantize(rng.standard_normal((n, side, side, cin)).astype(np.float32), dtype)
    cw = make_conv_weights(rng, cout, cin, 3)
    cwq = ConvWeights(w=quantize(cw.w, dtype), bn=cw.bn, bias=cw.bias)
    res = quantize(rng.standard_normal((n, side, side, cout)).astype(np.float32), dtype) if use_res else None
    want = _ref(x, cwq, 3, 1, act, res, False)
    atol, rtol = TOL[dtype]
    lib = ext.load()
    base32 = None
    for t32 in (36, 33):                 # a 32x32x16 implicit-GEMM tile that fits the layer: the order the KC = 64 halo2 tiles share
        try:
            base32, _ = run_conv_gpu(x, cwq, 3, 1, act, dtype, residual=res, in_pad=in_pad, out_pad=out_pad, tile=t32)
            break
        except ext.Y4Error:
            continue
    ran = 0
    for tile in range(1, lib.y4_conv_tile_count() + 1):
        cfg = (C.c_int32 * 6)()
        ext.check(lib.y4_conv_tile_desc(tile, cfg))
        if cfg[5] != 21:
            continue
        try:
            got, full = run_conv_gpu(x, cwq, 3, 1, act, dtype, residual=res, in_pad=in_pad, out_pad=out_pad, tile=tile)
        except ext.Y4Error as e:
            assert e.code == -22          # this band geometry / channel tile does not fit the layer: refused, not computed wrongly
            continue
        ran += 1
        err = np.abs(got - want)
        assert np.all(err <= atol + rtol * np.abs(want)), f"halo2 tile {tile}: max err {err.max():.3e} at {np.unravel_index(err.argmax(), err.shape)}"
        if cfg[4] == 128 and base32 is not None:
            assert np.array_equal(got, base32), f"halo2 tile {tile} differs from the 32x32x16 implicit-GEMM tile: {np.abs(got - base32).max():.3e}"
        again, _ = run_conv_gpu(x, cwq, 3, 1, act, dtype, residual=res, in_pad=in_pad, out_pad=out_pad, tile=tile)
        assert np.array_equal(got, again), f"halo2 tile {tile} is not repeatable"
        if out_pad[0]:
            assert np.all(full[..., :out_pad[0]] == -5.0)
        if out_pad[1]:
            assert np.all(full[..., out_pad[0] + cout:] == -5.0)
    assert ran >= 2, "fewer than two halo2 tiles fit this case"


@pytest.mark.parametrize("dtype", ["bf16", "f16"])
def test_pack_conv_frag32_layout(dtype):
    """y4_pack_conv_frag32 against a NumPy re-layout of the same packed matrix (include/yolo4hip.h): element e of lane l of k-step s of
    (32-channel block b, chunk c, tap t) = packed[ch(b, l & 31)][c][t][16 s + 8 (l >> 5) + e], with MFMA row R = 8 g + 4 h + j of a block
    <-> channel 32 b + 16 (g >> 1) + 8 h + 4 (g & 1) + j -- the layout conv_halo2_kernel's weight loads and the shared epilogue's channel
    order both rest on.  Raw 16-bit patterns are compared: a re-layout must not touch a bit."""
    import ctypes as C
    import torch
    from yolo4hip import ext
    lib = ext.load()
    did = ext.DTYPE_IDS[dtype]
    cout, cin = 200, 192                                   # cout_pad 256 (rows >= 200 zero), three 64-channel chunks
    rng = np.random.default_rng(3)
    w = torch.from_numpy((rng.standard_normal((cout, cin, 3, 3)) * 0.05).astype(np.float32)).to("cuda:0")
    cpad, nbytes = C.c_int32(), C.c_size_t()
    ext.check(lib.y4_packed_conv_bytes(did, cout, cin, 3, C.byref(cpad), C.byref(nbytes)))
    packed = torch.zeros(nbytes.value, dtype=torch.uint8, device="cuda:0")
    frag = torch.zeros(nbytes.value, dtype=torch.uint8, device="cuda:0")
    ext.check(lib.y4_pack_conv_weights(did, cout, cin, 3, ext.ptr(w), ext.ptr(packed), ext.stream_ptr()))
    ext.check(lib.y4_pack_conv_frag32(did, cout, cin, ext.ptr(packed), ext.ptr(frag), ext.stream_ptr()))
    torch.cuda.synchronize()
    nch = cin // 64
    p = packed.cpu().numpy().view(np.uint16).reshape(cpad.value, nch, 9, 64)
    f = frag.cpu().numpy().view(np.uint16).reshape(cpad.value // 32, nch, 9, 4, 64, 8)
    R = np.arange(32)
    g, h, j = R >> 3, (R >> 2) & 1, R & 3
    ch_of_row = 16 * (g >> 1) + 8 * h + 4 * (g & 1) + j    # channel within the block of MFMA row R
    want = np.empty_like(f)
    for b in range(cpad.value // 32):
        rows = p[32 * b + ch_of_row]                        # [R][chunk][tap][64]
        for s in range(4):
            for hk in range(2):
                # lanes hk*32 + R hold input channels 16 s + 8 hk .. +7
                want[b, :, :, s, hk * 32:(hk + 1) * 32, :] = rows[:, :, :, 16 * s + 8 * hk:16 * s + 8 * hk + 8].transpose(1, 2, 0, 3)
    assert np.array_equal(f, want)
    assert not f[200 // 32 + 1:].any()                      # blocks made of padding rows only
    with pytest.raises(ext.Y4Error):
        ext.check(lib.y4_pack_conv_frag32(ext.DTYPE_IDS["f32"], cout, cin, ext.ptr(packed), ext.ptr(frag), ext.stream_ptr()))
    with pytest.raises(ext.Y4Error):
        ext.check(lib.y4_pack_conv_frag32(did, cout, 96, ext.ptr(packed), ext.ptr(frag), ext.stream_ptr()))


def test_halo2_tiles_refuse_what_they_cannot_run():
    """A halo2 tile id on a 1x1 conv, a stride-2 conv, float32, Cin % 64 != 0 -- or without the fragment-ordered weights -- is refused
    (Y4_EINVAL), never mis-run."""
    import ctypes as C
    from yolo4hip import ext
    rng = np.random.default_rng(0)
    for k, stride, side, cin, cout, dtype in ((1, 1, 19, 128, 128, "bf16"), (3, 2, 38, 128, 128, "bf16"), (3, 1, 19, 128, 128, "f32"),
                                              (3, 1, 19, 32, 128, "bf16")):
        x = quantize(rng.standard_normal((1, side, side, cin)).astype(np.float32), dtype)
        cw = make_conv_weights(rng, cout, cin, k)
        with pytest.raises(ext.Y4Error) as e:
            run_conv_gpu(x, cw, k, stride, "mish", dtype, tile=55)
        assert e.value.code == -22
    import torch
    lib = ext.load()
    d = ext.y4_conv_desc()
    buf = torch.zeros(1 << 20, dtype=torch.uint8, device="cuda:0")
    d.dtype = ext.DTYPE_IDS["bf16"]; d.n, d.h, d.w, d.cin = 1, 19, 19, 128
    d.cout, d.ksize, d.stride, d.act = 128, 3, 1, 2
    d.in_cstride, d.out_cstride = 128, 128
    d.in_ = d.wt = d.scale = d.shift = d.out = buf.data_ptr(); d.tile = 55          # wt_frag stays NULL
    assert lib.y4_conv2d(C.byref(d), ext.stream_ptr()) == -22
    assert b"fragment-ordered" in lib.y4_last_error()


def test_halo_tiles_refuse_what_they_cannot_run():
    """A halo tile id on a 1x1 conv, a stride-2 conv, float32, or a map wider than the tile is refused (Y4_EINVAL), never mis-run."""
    from yolo4hip import ext
    rng = np.random.default_rng(0)
    for k, stride, side, cin, cout, dtype in ((1, 1, 19, 128, 128, "bf16"), (3, 2, 38, 128, 128, "bf16"), (3, 1, 19, 128, 128, "f32"),
                                              (3, 1, 19, 32, 128, "bf16")):
        x = quantize(rng.standard_normal((1, side, side, cin)).astype(np.float32), dtype)
        cw = make_conv_weights(rng, cout, cin, k)
        with pytest.raises(ext.Y4Error) as e:
            run_conv_gpu(x, cw, k, stride, "mish", dtype, tile=51)
        assert e.value.code == -22


@pytest.mark.parametrize("dtype", ["f32", "bf16"])
def test_splitk_fixup_sees_this_launch_partial_sums(dtype):
    """The split-K fix-up has no cache maintenance any more (round 4: agent-scope accesses + "my stores have completed" instead of
    __threadfence, conv_igemm_kernel.h): 25 launches each of two-, four- and eight-way splits on a scratch that is filled with NaNs
    before every launch -- a last arriver that read anything but this launch's partial sums would not reproduce the first result,
    which is held to the oracle."""
    from yolo4hip.weights import ConvWeights
    atol, rtol = TOL[dtype]
    rng = np.random.default_rng(77)
    for (k, cin, cout, side, tile) in [(3, 512, 512, 19, 148), (3, 256, 512, 19, 210), (1, 1024, 512, 19, 349), (3, 512, 1024, 13, 308)]:
        x = quantize(rng.standard_normal((1, side, side, cin)).astype(np.float32), dtype)
        cw = make_conv_weights(rng, cout, cin, k)
        cwq = ConvWeights(w=quantize(cw.w, dtype), bn=cw.bn)
        want = _ref(x, cwq, k, 1, "leaky", None, False)
        got, _ = run_conv_gpu(x, cwq, k, 1, "leaky", dtype, tile=tile, splitk_repeats=25)
        d = np.abs(got - want)
        assert np.all(d <= atol + rtol * np.abs(want)), f"tile {tile} {k}x{k} {cin}->{cout}: max err {d.max():.3e}"


@pytest.mark.parametrize("dtype", ["bf16", "f16"])
def test_mfma32_tiles_vs_oracle(dtype):
    """The 32x32x16-MFMA tiles (schedule code 32) against the same float64 reference and tolerance as every other tile: a 3x3
    conv with residual + Mish, a strided one and a 1x1 with a ragged pixel count and Cout = 255 (partial channel tile)."""
    import ctypes as C
    from yolo4hip import ext
    from yolo4hip.weights import ConvWeights
    lib = ext.load()
    tiles32 = []
    for tile in range(1, lib.y4_conv_tile_count() + 1):
        cfg = (C.c_int32 * 6)()
        ext.check(lib.y4_conv_tile_desc(tile, cfg))
        if cfg[5] == 32: tiles32.append(tile)
    assert len(tiles32) >= 3, tiles32
    atol, rtol = TOL[dtype]
    ran = 0
    for (k, stride, cin, cout, side, act, use_res) in [(3, 1, 128, 256, 19, "mish", True), (3, 2, 64, 128, 26, "leaky", False),
                                                       (1, 1, 512, 255, 13, "linear", False)]:
        rng = np.random.default_rng(cin + cout)
        x = quantize(rng.standard_normal((2, side, side, cin)).astype(np.float32), dtype)
        cw = make_conv_weights(rng, cout, cin, k, act != "linear")
        cwq = ConvWeights(w=quantize(cw.w, dtype), bn=cw.bn, bias=cw.bias)
        so = side // stride
        res = quantize(rng.standard_normal((2, so, so, cout)).astype(np.float32), dtype) if use_res else None
        want = _ref(x, cwq, k, stride, act, res, False)
        for tile in tiles32:
            try:
                got, _ = run_conv_gpu(x, cwq, k, stride, act, dtype, residual=res, tile=tile)
            except ext.Y4Error as e:
                assert e.code == -22
                continue
            err = np.abs(got - want)
            assert np.all(err <= atol + rtol * np.abs(want)), f"tile {tile} {k}x{k} {cin}->{cout}: max err {err.max():.3e}"
            ran += 1
    assert ran >= 6


@pytest.mark.parametrize("dtype", ["f32", "bf16", "f16"])
def test_deep_ring_tiles_short_and_long_k(dtype):
    """The ring tiles with 3..7 LDS stages (conv_tiles.h; the deep ones are what a single image's latency-bound K loops run on)
    with FEWER K-tiles than stages (1x1 convs over 64 / 128 / 192 channels at bf16: 1, 2, 3 K-tiles), exactly as many, and many
    more: bit-identical to the two-stage 64x64 tile (same K order), which is held to the oracle; also as split-K bases."""
    import ctypes as C
    from yolo4hip import ext
    from yolo4hip.weights import ConvWeights
    lib = ext.load()
    deep = []
    for tile in range(1, lib.y4_conv_tile_count() + 1):
        cfg = (C.c_int32 * 6)()
        ext.check(lib.y4_conv_tile_desc(tile, cfg))
        if 3 <= cfg[5] <= 7: deep.append((tile, cfg[5]))
    assert max(n for _, n in deep) >= 6 and len(deep) >= 6, deep
    atol, rtol = TOL[dtype]
    ran = ran_split = 0
    for (k, cin, cout, side, act, use_res) in [(1, 64, 128, 19, "mish", False), (1, 128, 64, 13, "leaky", False), (1, 192, 128, 19, "mish", True),
                                                (1, 384, 255, 13, "linear", False), (3, 128, 128, 19, "mish", True), (3, 512, 256, 13, "leaky", False)]:
        rng = np.random.default_rng(cin + cout + k)
        x = quantize(rng.standard_normal((1, side, side, cin)).astype(np.float32), dtype)
        cw = make_conv_weights(rng, cout, cin, k, act != "linear")
        cwq = ConvWeights(w=quantize(cw.w, dtype), bn=cw.bn, bias=cw.bias)
        res = quantize(rng.standard_normal((1, side, side, cout)).astype(np.float32), dtype) if use_res else None
        want = _ref(x, cwq, k, 1, act, res, False)
        base, _ = run_conv_gpu(x, cwq, k, 1, act, dtype, residual=res, tile=10)
        d = np.abs(base - want)
        assert np.all(d <= atol + rtol * np.abs(want)), f"tile 10 {k}x{k} {cin}->{cout}: max err {d.max():.3e}"
        for tile, nst in deep:
            try:
                got, _ = run_conv_gpu(x, cwq, k, 1, act, dtype, residual=res, tile=tile)
            except ext.Y4Error as err:
                assert err.code == -22      # (64-byte K rows against this cin, or a tile that is not built for float32)
                continue
            assert np.array_equal(got, base), f"tile {tile} ({nst} stages) {k}x{k} {cin}->{cout}: {np.abs(got - base).max()}"
            ran += 1
            if nst >= 5:
                try:
                    got, _ = run_conv_gpu(x, cwq, k, 1, act, dtype, residual=res, tile=tile + 100)
                except ext.Y4Error as err:
                    assert err.code == -22
                    continue
                d = np.abs(got - want)
                assert np.all(d <= atol + rtol * np.abs(want)), f"tile {tile}+100 {k}x{k} {cin}->{cout}: max err {d.max():.3e}"
                ran_split += 1
    assert ran >= 30 and ran_split >= 6, (ran, ran_split)


@pytest.mark.parametrize("dtype", ["f32", "bf16", "f16"])
def test_splitk_tiles_vs_oracle(dtype):
    """Split-K tile ids (base + 100 e: the K loop split 2^e ways, the last split to arrive adds the partial sums and runs the
    epilogue; the latency schedules of batch 1) against the same float64-free reference and tolerance as every other tile --
    they sum in another fp32 order, so they are held to the oracle, not to bit-identity: the deep 19^2 3x3 shapes, a strided one
    with a residual, a 1x1 head with Cout = 255 and float32 output, a ragged pixel count; every launch is repeated on its scratch
    (run_conv_gpu) to prove the tile counters return to zero.  Ids that cannot split (too few K-tiles, fused tiles) are refused."""
    import ctypes as C
    from yolo4hip import ext
    from yolo4hip.weights import ConvWeights
    lib = ext.load()
    atol, rtol = TOL[dtype]
    bases = [8, 10, 12, 16, 3] if dtype != "f32" else [8, 10, 3]
    ran = 0
    for (k, stride, cin, cout, side, act, use_res, out_f32) in [(3, 1, 512, 1024, 19, "leaky", False, False),
                                                                 (3, 2, 256, 512, 38, "mish", False, False),
                                                                 (3, 1, 256, 256, 19, "mish", True, False),
                                                                 (1, 1, 1024, 255, 19, "linear", False, True),
                                                                 (1, 1, 512, 256, 13, "leaky", False, False)]:
        rng = np.random.default_rng(cin + cout + k)
        x = quantize(rng.standard_normal((1, side, side, cin)).astype(np.float32), dtype)
        cw = make_conv_weights(rng, cout, cin, k, act != "linear")
        cwq = ConvWeights(w=quantize(cw.w, dtype), bn=cw.bn, bias=cw.bias)
        so = side // stride
        res = quantize(rng.standard_normal((1, so, so, cout)).astype(np.float32), dtype) if use_res else None
        want = _ref(x, cwq, k, stride, act, res, False)
        for base in bases:
            for e in (1, 2, 3):
                try:
                    got, _ = run_conv_gpu(x, cwq, k, stride, act, dtype, residual=res, out_f32=out_f32, tile=base + 100 * e)
                except ext.Y4Error as err:
                    assert err.code == -22
                    continue
                d = np.abs(got - want)
                assert np.all(d <= atol + rtol * np.abs(want)), f"tile {base}+{100 * e} {k}x{k} {cin}->{cout}: max err {d.max():.3e}"
                ran += 1
    assert ran >= 30, ran
    # refused: a split of a fused / phased tile id, a 16-way split, a split wider than the K-tiles allow
    rng = np.random.default_rng(1)
    x = quantize(rng.standard_normal((1, 13, 13, 64)).astype(np.float32), dtype)
    cw = make_conv_weights(rng, 64, 64, 1, True)
    for bad in ([430] if dtype == "f32" else [119 + 11, 133, 430]) + [408, 308]:     # 30: staggered; 33: 32x32x16; e = 4; K too short
        with pytest.raises(ext.Y4Error):
            run_conv_gpu(x, cw, 1, 1, "mish", dtype, tile=bad)


@pytest.mark.parametrize("dtype", ["bf16", "f16"])
def test_phased_tiles(dtype):
    """The phased kernel (conv_p8_kernel.h, schedule codes 8: staggered wave groups, 9: software-pipelined; and conv_l12_kernel.h,
    code 10: four producer waves stage, eight consumer waves multiply; region-wise LDS recycling, counted vmcnt) against the float64 reference AND bit for bit against the plain 192x256 kernel: 3x3 with residual
    + Mish, stride 2 (asymmetric padding), 1x1 with a ragged pixel count and Cout = 255, short K (one and two K-tiles: the
    pipeline is then mostly prologue / tail), channel-slice views, 2x2 replicated store, float32 store."""
    import ctypes as C
    from yolo4hip import ext
    from yolo4hip.weights import ConvWeights
    lib = ext.load()
    tiles = {}
    for tile in range(1, lib.y4_conv_tile_count() + 1):
        cfg = (C.c_int32 * 6)()
        ext.check(lib.y4_conv_tile_desc(tile, cfg))
        tiles[tile] = tuple(cfg)
    phased = [t for t, c in tiles.items() if c[5] in (8, 9, 10)]
    assert len(phased) >= 4
    plain16 = next(t for t, c in tiles.items() if c[:2] == (192, 256) and c[5] == 2)
    atol, rtol = TOL[dtype]
    cases = [  # k, stride, cin, cout, n, side, act, residual, kwargs
        (3, 1, 128, 256, 3, 19, "mish", True, {}),
        (3, 2, 64, 256, 2, 26, "leaky", False, {}),
        (1, 1, 512, 255, 5, 13, "linear", False, dict(out_f32=True)),
        (1, 1, 64, 512, 2, 24, "leaky", False, {}),
        (1, 1, 128, 256, 1, 20, "mish", False, dict(upsample=True, in_pad=(64, 8), out_pad=(16, 24))),
        (3, 1, 256, 512, 1, 38, "leaky", False, {}),
    ]
    for (k, stride, cin, cout, n, side, act, use_res, kw) in cases:
        rng = np.random.default_rng(cin + cout + side)
        x = quantize(rng.standard_normal((n, side, side, cin)).astype(np.float32), dtype)
        cw = make_conv_weights(rng, cout, cin, k, act != "linear")
        cwq = ConvWeights(w=quantize(cw.w, dtype), bn=cw.bn, bias=cw.bias)
        so = side // stride
        res = quantize(rng.standard_normal((n, so, so, cout)).astype(np.float32), dtype) if use_res else None
        want = _ref(x, cwq, k, stride, act, res, kw.get("upsample", False))
        ref16, _ = run_conv_gpu(x, cwq, k, stride, act, dtype, residual=res, tile=plain16, **kw)
        for tile in phased:
            got, full = run_conv_gpu(x, cwq, k, stride, act, dtype, residual=res, tile=tile, **kw)
            a, r = (1e-4, 1e-4) if kw.get("out_f32") else (atol, rtol)
            err = np.abs(got - want)
            assert np.all(err <= a + r * np.abs(want)), f"tile {tile} {k}x{k} {cin}->{cout}: max err {err.max():.3e}"
            same = ref16
            assert np.array_equal(got, same), f"tile {tile} {k}x{k} {cin}->{cout}: differs from the plain kernel by {np.abs(got - same).max():.3e}"
            if kw.get("out_pad"):
                lo, hi = kw["out_pad"]
                assert np.all(full[..., :lo] == -5.0) and np.all(full[..., lo + (cout + 7) // 8 * 8:] == -5.0)


def test_conv_rejects_bad_shapes():
    import ctypes as C
    from yolo4hip import ext
    lib = ext.load()
    d = ext.y4_conv_desc()
    assert lib.y4_conv2d(C.byref(d), None) == -22
    assert b"null" in lib.y4_last_error()
