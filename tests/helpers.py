"""Shared test helpers (host side only; the oracle is imported by the tests themselves)."""
import ctypes as C
import os

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
PKG_DIR = os.path.join(ROOT, "yolo-v4-tf.keras_amd")
GOLDEN = os.path.join(ROOT, "tests", "golden")
CLASS_DIR = os.path.join(PKG_DIR, "class_names")


def bf16_round(x):
    """float32 -> nearest-even bfloat16 -> float32 (what the HIP path stores)."""
    u = np.ascontiguousarray(x, dtype=np.float32).view(np.uint32).astype(np.uint64)
    u = (u + 0x7FFF + ((u >> 16) & 1)) & 0xFFFF0000
    return u.astype(np.uint32).view(np.float32).reshape(np.shape(x))


def quantize(x, dtype):
    if dtype == "f32":
        return np.ascontiguousarray(x, dtype=np.float32)
    if dtype == "bf16":
        return bf16_round(x)
    return np.asarray(x, dtype=np.float32).astype(np.float16).astype(np.float32)


def torch_dtype(dtype):
    import torch
    return {"f32": torch.float32, "bf16": torch.bfloat16, "f16": torch.float16}[dtype]


def run_conv_gpu(x, cw, k, stride, act, dtype, residual=None, upsample=False, out_f32=False, tile=0,
                 in_pad=(0, 0), out_pad=(0, 0), splitk_repeats=1):
    """Run y4_conv2d on NHWC float32 numpy `x` (already representable in `dtype`).
    in_pad/out_pad = (channels before, channels after) of extra garbage around the view, to exercise
    channel-slice reads/stores.  Returns float32 NHWC output (the slice only) and the full out buffer."""
    import torch
    from yolo4hip import ext
    lib = ext.load()
    dev = "cuda:0"
    td = torch_dtype(dtype)
    did = ext.DTYPE_IDS[dtype]
    n, h, w, cin = x.shape
    cout = cw.w.shape[0]
    ho, wo = h // stride, w // stride
    oh, ow = (2 * ho, 2 * wo) if upsample else (ho, wo)
    in_cs = in_pad[0] + cin + in_pad[1]
    xin = torch.full((n, h, w, in_cs), 7.0, dtype=td, device=dev)
    xin[..., in_pad[0]:in_pad[0] + cin] = torch.from_numpy(x).to(dev).to(td)
    cstore = (cout + 7) // 8 * 8
    out_cs = out_pad[0] + cstore + out_pad[1]
    otd = torch.float32 if out_f32 else td
    out = torch.full((n, oh, ow, out_cs), -5.0, dtype=otd, device=dev)
    cpad, nbytes = C.c_int32(), C.c_size_t()
    ext.check(lib.y4_packed_conv_bytes(did, cout, cin, k, C.byref(cpad), C.byref(nbytes)))
    packed = torch.empty(nbytes.value, dtype=torch.uint8, device=dev)
    w_dev = torch.from_numpy(np.ascontiguousarray(cw.w)).to(dev)
    ext.check(lib.y4_pack_conv_weights(did, cout, cin, k, ext.ptr(w_dev), ext.ptr(packed), ext.stream_ptr()))
    scale, shift = cw.scale_shift()
    sc = torch.zeros(cpad.value, dtype=torch.float32, device=dev); sc[:cout] = torch.from_numpy(scale).to(dev)
    sh = torch.zeros(cpad.value, dtype=torch.float32, device=dev); sh[:cout] = torch.from_numpy(shift).to(dev)
    d = ext.y4_conv_desc()
    d.dtype = did; d.n, d.h, d.w, d.cin = n, h, w, cin
    d.cout, d.ksize, d.stride, d.act = cout, k, stride, {"mish": 2, "leaky": 1, "linear": 0, None: 0}[act]
    d.upsample, d.out_f32 = int(upsample), int(out_f32)
    d.in_cstride, d.in_coff = in_cs, in_pad[0]
    d.out_cstride, d.out_coff = out_cs, out_pad[0]
    d.in_ = xin.data_ptr(); d.wt = packed.data_ptr(); d.scale = sc.data_ptr(); d.shift = sh.data_ptr()
    d.out = out.data_ptr(); d.tile = tile
    frag = None
    if k == 3 and dtype != "f32" and cin % 64 == 0 and 1 <= tile % 100 <= lib.y4_conv_tile_count():
        cfg = (C.c_int32 * 6)()
        ext.check(lib.y4_conv_tile_desc(tile % 100, cfg))
        if cfg[5] == 21:                 # a halo2 tile: the weights once more in MFMA-fragment order
            frag = torch.empty(nbytes.value, dtype=torch.uint8, device=dev)
            ext.check(lib.y4_pack_conv_frag32(did, cout, cin, ext.ptr(packed), ext.ptr(frag), ext.stream_ptr()))
            d.wt_frag = frag.data_ptr()
    res_t = None
    if residual is not None:
        res_t = torch.from_numpy(residual).to(dev).to(td).contiguous()
        d.res = res_t.data_ptr(); d.res_cstride = residual.shape[-1]; d.res_coff = 0
    ws = None
    if tile >= 100:                      # split-K tile id: counters (zero before the first use) + partial sums
        ws = torch.zeros(16 * 1024 + 32 * 1024 * 1024, dtype=torch.uint8, device=dev)
        d.splitk_ws = ws.data_ptr(); d.splitk_ws_bytes = ws.numel()
    if ws is not None: ws[16 * 1024:].fill_(0xFF)       # the partial sums' scratch starts as NaNs: a stale read shows
    ext.check(lib.y4_conv2d(C.byref(d), ext.stream_ptr()))
    if ws is not None:                   # more launches on the same scratch: the counters were left at zero, and the last arriver
        first = out.clone()              # must read THIS launch's partial sums (the scratch is poisoned in between)
        for _ in range(splitk_repeats):
            out.fill_(-5.0)
            ws[16 * 1024:].fill_(0xFF)
            ext.check(lib.y4_conv2d(C.byref(d), ext.stream_ptr()))
            torch.cuda.synchronize()
            assert torch.equal(first, out), "a split-K launch is not repeatable on its own scratch"
        assert int(ws[:16 * 1024].view(torch.int32).abs().sum()) == 0, "split-K tile counters not back at zero"
    torch.cuda.synchronize()
    full = out.float().cpu().numpy()
    return full[..., out_pad[0]:out_pad[0] + cout], full


def make_conv_weights(rng, cout, cin, k, bn=True):
    from yolo4hip.weights import ConvWeights
    w = (rng.standard_normal((cout, cin, k, k)) / np.sqrt(cin * k * k)).astype(np.float32)
    if bn:
        b = np.stack([rng.standard_normal(cout) * 0.2, 1.0 + 0.3 * rng.standard_normal(cout),
                      0.2 * rng.standard_normal(cout), 0.5 + rng.random(cout)]).astype(np.float32)
        return ConvWeights(w=w, bn=b)
    return ConvWeights(w=w, bias=(rng.standard_normal(cout) * 0.5).astype(np.float32))


def randomize_bn(ws, seed):
    """Function-preserving re-parametrisation of a weight set with NON-TRIVIAL BatchNorm statistics: every BN
    conv gets random mean (+-0.5), var (0.3..3) and gamma (0.5..1.5); the kernel and beta are adjusted so that
    the layer computes the same function in exact arithmetic (the synthetic net stays well conditioned), while
    `scale = gamma*rsqrt(var+eps)`, `shift = beta - mean*scale` (reference utils.py:28-31, custom_layers.py:26)
    now depend on all four rows.  A sign or row-order slip in the device-side fold then shows up at O(1)."""
    from yolo4hip.weights import BN_EPS, ConvWeights
    rng = np.random.default_rng([seed, 0xB17])
    out = []
    for cw in ws:
        if cw.bn is None:
            out.append(cw)
            continue
        beta, gamma, mean, var = cw.bn.astype(np.float64)
        cout = beta.size
        scale0 = gamma / np.sqrt(var + BN_EPS)
        shift0 = beta - mean * scale0
        g2 = rng.uniform(0.5, 1.5, cout)
        v2 = rng.uniform(0.3, 3.0, cout)
        m2 = rng.uniform(-0.5, 0.5, cout)
        scale2 = g2 / np.sqrt(v2 + BN_EPS)
        w2 = (cw.w.astype(np.float64) * (scale0 / scale2)[:, None, None, None]).astype(np.float32)
        beta2 = shift0 + m2 * scale2
        bn2 = np.stack([beta2, g2, m2, v2]).astype(np.float32)
        out.append(ConvWeights(w=w2, bn=bn2))
    return out


def detection_agreement(kept, classes, scores, boxes, valid, ri, rc, rs, rb, rv):
    """Per-image agreement of two NMS results matched by (box index, class): fraction of the reference's
    detections also present, max |score delta| and max |box delta| over the matched ones."""
    got = {(int(kept[j]), int(classes[j])): j for j in range(int(valid))}
    ref = {(int(ri[j]), int(rc[j])): j for j in range(int(rv))}
    common = set(got) & set(ref)
    ds = max((abs(float(scores[got[k]]) - float(rs[ref[k]])) for k in common), default=0.0)
    db = max((float(np.abs(boxes[got[k]] - rb[ref[k]]).max()) for k in common), default=0.0)
    return (len(common) / max(len(ref), 1)), ds, db


def score_delta_quantile(kept, classes, scores, valid, ri, rc, rs, rv, q=0.9):
    """The q-quantile of |score delta| over the detections two NMS results share (matched by box index and class): the bulk of
    the distribution, where `detection_agreement`'s maximum is its tail."""
    got = {(int(kept[j]), int(classes[j])): j for j in range(int(valid))}
    ref = {(int(ri[j]), int(rc[j])): j for j in range(int(rv))}
    d = [abs(float(scores[got[k]]) - float(rs[ref[k]])) for k in set(got) & set(ref)]
    return float(np.quantile(d, q)) if d else 0.0


def shift_objectness(ws, num_classes, delta):
    """A copy of a weight set whose three head convs (93, 101, 109: bias, no BN; custom_layers.py:141-198) have `delta` added to
    the bias of their objectness channels (channel 4 of each anchor's 5 + C block): with delta < 0 fewer boxes pass the score
    threshold -- the `valid < 100` regime of CombinedNMS (tests/golden/make_ref_sparse_fixture.py)."""
    from yolo4hip.weights import ConvWeights
    out = list(ws)
    for idx in (93, 101, 109):
        cw = out[idx]
        assert cw.bias is not None and cw.bias.shape[0] == 3 * (5 + num_classes), (idx, None if cw.bias is None else cw.bias.shape)
        b = cw.bias.copy()
        b[4::5 + num_classes] += np.float32(delta)
        out[idx] = ConvWeights(w=cw.w, bn=cw.bn, bias=b)
    return out


# ---------------------------------------------------------------- determinism helpers (tests/test_gpu_determinism.py)
def tap_snapshot(eng, n):
    """Every tensor the forward that just ran left behind, as DEVICE tensors in graph order: each materialised conv output
    (y4_get_conv_output; convs inside a fused kernel are skipped), the three float32 heads, and the decode + NMS outputs."""
    import torch
    from yolo4hip import ext
    if not hasattr(eng, "_lt"):
        eng._lt = eng.layer_table()
    snap = {}
    if not getattr(eng, "alias_workspace", False):
        for i, lt in enumerate(eng._lt):
            side = lt["out_side"] * (2 if i in (78, 85) else 1)
            out = torch.empty((n, side, side, lt["cout"]), dtype=torch.float32, device=eng.device)
            if eng.lib.y4_get_conv_output(eng.handle, i, n, ext.ptr(out), out.numel(), ext.stream_ptr()) == 0:
                snap[f"conv{i}"] = out
    for k, h in enumerate(eng.heads_device(n)):
        snap[f"head{k}"] = h
    for name, t in zip(("boxes", "scores", "classes", "valid", "kept"), eng.decode_nms_device(n)):
        snap[name] = t
    torch.cuda.synchronize()
    return snap


def first_tap_difference(a, b, eng=None):
    """None when the two snapshots are bit-identical, else a one-line description of the FIRST differing tensor in graph order
    (the origin of a divergence): name, layer shape, how many elements differ, where, and by how much."""
    import torch
    for name in a:
        if name not in b or torch.equal(a[name], b[name]):
            continue
        x, y = a[name].float(), b[name].float()
        d = x != y
        idx = d.nonzero()
        where = idx[:4].cpu().numpy().tolist()
        msg = f"{name}: {int(d.sum())} of {d.numel()} elements differ, first at {where}, max |delta| {float((x - y).abs().max()):.3e}"
        if idx.shape[1] == 4:
            ii = idx.cpu().numpy()
            msg += (f", images {sorted(set(ii[:, 0].tolist()))}, rows {int(ii[:, 1].min())}..{int(ii[:, 1].max())}, cols "
                    f"{int(ii[:, 2].min())}..{int(ii[:, 2].max())}, channels {int(ii[:, 3].min())}..{int(ii[:, 3].max())}")
        if eng is not None and name.startswith("conv"):
            lt = eng._lt[int(name[4:])]
            msg += f" [k{lt['ksize']} s{lt['stride']} {lt['cin']}->{lt['cout']} @{lt['out_side']}]"
        return msg
    return None
