"""GPU parity of the whole hot path (y4_forward / y4_decode_nms through the C ABI) against the oracle
(oracle/forward.py, oracle/decode_nms.py), on seeded synthetic weights and images.

north_star tolerance: box coords / scores within 1e-3 (fp32), identical kept-box indices after NMS.
The raw-head tolerance used here for fp32 is 2e-3 absolute on logits of O(1..10) magnitude: the HIP
path and oneDNN sum the same fp32 products in different orders through 110 layers.
For bf16/fp16 the heads are compared with a loose bound (documented per test): identical kept indices
are not claimed at 16-bit precision (SURVEY.md §7 hard part iv).
"""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu


def _setup(size, ncls, n, dtype, seed=0):
    from yolo4hip import weights as W
    from yolo4hip.config import make_config
    from yolo4hip.engine import Engine
    from yolo4hip.plan import build_plan
    cfg = make_config(size)
    plan = build_plan(size, ncls)
    ws = W.synth_weights(plan, seed)
    imgs = W.synth_images(n, size, seed)
    eng = Engine(ncls, cfg, max_batch=n, dtype=dtype)
    eng.load_weight_blob(W.flatten(ws))
    return cfg, plan, ws, imgs, eng


def _same_detections(kept, classes, scores, boxes, valid, ri, rc, rs, rb, rv, score_thr, tol=1e-3):
    got = {(int(kept[j]), int(classes[j])): j for j in range(valid)}
    ref = {(int(ri[j]), int(rc[j])): j for j in range(rv)}
    cutoff = min(rs[rv - 1], scores[valid - 1]) if rv and valid else 0.0
    for key in set(got) ^ set(ref):
        sc = scores[got[key]] if key in got else rs[ref[key]]
        near = abs(sc - score_thr) < tol or (max(rv, valid) == len(rs) and abs(sc - cutoff) < tol)
        assert near, f"detection {key} (score {sc:.6f}) present on one side only and not near a threshold"
    common = sorted(set(got) & set(ref), key=lambda k_: got[k_])
    assert len(common) >= 0.95 * rv
    for key in common:
        j, r = got[key], ref[key]
        assert abs(scores[j] - rs[r]) < tol, (key, scores[j], rs[r])
        assert np.abs(boxes[j] - rb[r]).max() < tol, (key, boxes[j], rb[r])
    ref_scores_in_got_order = np.array([rs[ref[k_]] for k_ in common])
    assert np.all(np.diff(ref_scores_in_got_order) < tol), "rank order differs beyond near-ties"
    assert np.all(np.diff(scores[:valid]) <= 0)


def test_plan_matches_python_plan():
    """The C++ plan (csrc/runtime.hip) and the Python plan (yolo4hip/plan.py) are independent statements
    of reference custom_layers.py:100-198; they must agree row by row."""
    from yolo4hip.config import make_config
    from yolo4hip.engine import Engine
    from yolo4hip.plan import build_plan
    for size, ncls in ((416, 80), (608, 3)):
        plan = build_plan(size, ncls)
        eng = Engine(ncls, make_config(size), max_batch=1, dtype="bf16")
        table = eng.layer_table()
        assert len(table) == 110
        off = 0
        for row, c in zip(table, plan.convs):
            assert (row["ksize"], row["stride"], row["cin"], row["cout"], row["act"], row["has_bn"],
                    row["in_side"], row["out_side"]) == (c.k, c.s, c.cin, c.cout, c.act, int(c.bn), c.in_side,
                                                         c.out_side), row
            assert row["weight_offset"] == off
            off += (4 if c.bn else 1) * c.cout + c.n_weights
        assert eng.flops_per_image == plan.flops_per_image
        assert eng.num_boxes == plan.num_boxes
        assert eng.weight_floats == plan.n_params
        eng.close()


@pytest.mark.parametrize("size,ncls,n", [(416, 3, 2), (608, 80, 1)])
def test_fp32_forward_and_nms_parity(size, ncls, n):
    """Config 2 of BASELINE.json (608x608 batch 1 fp32, 80 classes) and the 416/3-class shape.

    A randomly initialised 110-layer net amplifies rounding noise (measured: the CPU oracle evaluated in
    fp32 differs from the same oracle in fp64 by up to ~1e-3 on the head logits, scripts/err_growth.py),
    so "equal to the fp32 oracle" is only meaningful up to that noise.  Checked here:
      1. every tapped layer and head: the HIP fp32 result is as close to the fp64 evaluation as the
         oracle's own fp32 evaluation is (factor 2 + 1e-5 slack) -- the first failing layer is named;
      2. heads within 3e-3 abs of the fp32 oracle; final boxes/scores within 1e-3 (north_star);
      3. the kept (box index, class) SET is identical to the oracle's, except for detections whose score is
         within the noise (1e-3) of the score threshold or of the rank-100 cut-off; rank order may differ
         only between detections whose scores are within 1e-3 of each other (near-ties swap under noise);
      4. the decode+NMS kernels on the ORACLE's heads: decisions bit-identical, values to an ulp.
    """
    import torch
    from oracle import forward as OF, decode_nms as OD
    cfg, plan, ws, imgs, eng = _setup(size, ncls, n, "f32")
    taps_idx = [0, 1, 2, 3, 4, 5, 6, 7, 8, 14, 17, 35, 37, 58, 71, 74, 77, 78, 84, 85, 91, 92, 99, 107, 108]
    ref_heads, taps = OF.yolo_model_forward(imgs, ws, ncls, collect=taps_idx)
    h64, taps64 = OF.yolo_model_forward(imgs, ws, ncls, dtype=torch.float64, collect=taps_idx)
    heads = eng.forward_heads(imgs)
    for idx in taps_idx:
        got = eng.conv_output(idx, n)
        want = taps.get(("add", idx), taps[idx])     # residual convs store conv + Add (fused epilogue)
        exact = taps64.get(("add", idx), taps64[idx])
        if idx in (78, 85):                          # these convs store the 2x nearest-upsampled tensor
            want = want.repeat(2, axis=1).repeat(2, axis=2)
            exact = exact.repeat(2, axis=1).repeat(2, axis=2)
        e_gpu, e_cpu = np.abs(got - exact).max(), np.abs(want - exact).max()
        assert e_gpu <= 2 * e_cpu + 1e-5, f"conv {idx}: HIP err vs fp64 {e_gpu:.3e}, oracle fp32 err {e_cpu:.3e}"
    for i, (a, b, c) in enumerate(zip(heads, ref_heads, h64)):
        assert a.shape == b.shape and a.dtype == np.float32
        e_gpu, e_cpu = np.abs(a - c).max(), np.abs(b - c).max()
        assert e_gpu <= 2 * e_cpu + 1e-5, f"head {i}: HIP err vs fp64 {e_gpu:.3e}, oracle fp32 err {e_cpu:.3e}"
        assert np.abs(a - b).max() < 3e-3, f"head {i}: max abs err vs fp32 oracle {np.abs(a - b).max():.3e}"
    boxes, scores, classes, valid, kept = eng.predict(imgs, with_indices=True)
    rb, rs, rc, rv, ri = OD.inference_from_heads(ref_heads, ncls, cfg["anchors"], cfg["xyscale"], size)
    assert boxes.shape == (n, 100, 4) and scores.shape == (n, 100) and classes.shape == (n, 100)
    assert valid.dtype == np.int32 and valid.shape == (n,)
    for b in range(n):
        _same_detections(kept[b], classes[b], scores[b], boxes[b], valid[b], ri[b], rc[b], rs[b], rb[b], rv[b],
                         cfg["score_threshold"])
    eng.set_heads(ref_heads)
    b2, s2, c2, v2, k2 = [o.cpu().numpy() for o in eng.decode_nms_device(n)]
    assert np.array_equal(v2, rv) and np.array_equal(k2, ri) and np.array_equal(c2, rc)
    assert np.abs(b2 - rb).max() < 1e-5 and np.abs(s2 - rs).max() < 1e-6
    eng.close()


# (bounds = measured + ~15 %, VERDICT r4 item 4: bf16 99.9 % quantile 0.27-0.33 and 93-94 % matched at 416 / 3 classes / batch 2,
#  fp16 0.034-0.040 and 98-99 %; the bulk / tail split of the score deltas is in tests/test_gpu_parity_full.py)
@pytest.mark.parametrize("dtype,tol,min_common", [("bf16", 0.40, 0.86), ("f16", 0.049, 0.96)])
def test_16bit_forward_close_to_fp32_oracle(dtype, tol, min_common):
    """16-bit storage: heads stay close to the fp32 oracle.  Bounds = measured on MI355X + ~15 % (bf16: 99.9 % quantile of
    |logit error| 0.27-0.33, mean 0.05-0.06, 92-96 % of detections matched by (box, class); fp16: 0.034-0.042, 0.0064-0.0075,
    98-99 %); the budget of 110 layers of 8-/11-bit mantissa rounding on logits of std ~1.3."""
    from oracle import forward as OF, decode_nms as OD
    size, ncls, n = 416, 3, 2
    cfg, plan, ws, imgs, eng = _setup(size, ncls, n, dtype)
    ref_heads = OF.yolo_model_forward(imgs, ws, ncls)
    heads = eng.forward_heads(imgs)
    measured = []
    for a, b in zip(heads, ref_heads):
        err = np.abs(a - b)
        assert np.isfinite(a).all()
        measured.append((float(err.mean()), float(np.quantile(err, 0.999))))
        assert err.mean() < tol / 5 and np.quantile(err, 0.999) < tol, (err.mean(), err.max())
    boxes, scores, classes, valid, kept = eng.predict(imgs, with_indices=True)
    rb, rs, rc, rv, ri = OD.inference_from_heads(ref_heads, ncls, cfg["anchors"], cfg["xyscale"], size)
    fracs = []
    for b in range(n):
        common = len(set(zip(kept[b, :valid[b]].tolist(), classes[b, :valid[b]].tolist())) &
                     set(zip(ri[b, :rv[b]].tolist(), rc[b, :rv[b]].tolist())))
        fracs.append(common / max(int(rv[b]), 1))
        assert common >= min_common * rv[b], (common, rv[b])
    try:
        import json, os
        with open(os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "gpurun_out", "parity_measured.jsonl"), "a") as f:
            f.write(json.dumps({"test": f"416_3_{dtype}_b2", "head_err_mean_q999": measured, "matched": fracs}) + "\n")
    except OSError:
        pass
    eng.close()


def test_batch_rows_independent_of_batch_size():
    """Images are independent units: image i of a batch of 3 equals image i run alone."""
    cfg, plan, ws, imgs, eng = _setup(160, 3, 3, "bf16", seed=3)
    full = eng.predict(imgs, with_indices=True)
    for i in range(3):
        one = eng.predict(imgs[i:i + 1], with_indices=True)
        for a, b in zip(full, one):
            assert np.array_equal(a[i:i + 1], b)
    eng.close()


@pytest.mark.parametrize("dtype,size,ncls,n", [("bf16", 416, 80, 3), ("f32", 160, 3, 2), ("f16", 608, 80, 1)])
def test_objectness_side_array_changes_nothing(dtype, size, ncls, n):
    """Round 4: the head convs also leave their cells' objectness logits in a dense side array that decode's screen reads instead
    of the cells (kernels.h: ConvObjDesc).  They are the stored logits themselves, so the detections of a forward pass (screen
    from the side array) equal those of the same raw heads loaded back through y4_set_heads (which voids the array: screen from
    the cells) bit for bit -- with the fusions on (two heads are LDS-pair tails at the 16-bit dtypes) and off."""
    cfg, plan, ws, imgs, eng = _setup(size, ncls, n, dtype, seed=5)
    for fused in (False, True):
        if dtype != "f32":
            eng.set_stem_fusion(fused); eng.set_chain_fusion(fused)
        a = eng.predict(imgs, with_indices=True)
        heads = [h.cpu().numpy() for h in eng.heads_device(n)]
        eng.set_heads(heads)
        b = [o.cpu().numpy() for o in eng.decode_nms_device(n)]
        assert a[3].sum() > 0
        for x, y in zip(a, b):
            assert np.array_equal(np.asarray(x), y)
    eng.close()


def test_full_size_properties():
    """BASELINE.json's headline configuration (608x608, 80 classes, batch 32, bf16, fusions on) through size-independent
    properties: the step is deterministic (two runs bit-equal), every image of the batch equals that image run alone
    (images are independent units, so any batch sharding over ranks gives the same rows), the fused schedule equals the
    plain one bit for bit, detections are well-formed (scores descending and above the threshold, boxes ordered,
    padding zeroed, kept indices unique per class), and a batch of all-zero images matches between batch and single."""
    import torch
    size, ncls, n = 608, 80, 32
    cfg, plan, ws, imgs, eng = _setup(size, ncls, n, "bf16", seed=7)
    imgs[5] = 0.0
    dev = torch.from_numpy(imgs).to(eng.device)
    plain = [o.cpu().numpy() for o in eng.predict_device(dev)]
    eng.set_stem_fusion(True)
    assert eng.set_chain_fusion(True) == 26      # 25 runs + the alternative run of the 152^2 stage (conv_chain.h CFG 4)
    assert eng.set_stage_fusion(True)
    run1 = [o.cpu().numpy() for o in eng.predict_device(dev)]
    run2 = [o.cpu().numpy() for o in eng.predict_device(dev)]
    for a, b, c in zip(run1, run2, plain):
        assert np.array_equal(a, b) and np.array_equal(a, c)
    boxes, scores, classes, valid, kept = run1
    thr = cfg["score_threshold"]
    assert valid.min() >= 0 and valid.max() <= 100 and valid.sum() > 0
    for b in range(n):
        v = int(valid[b])
        assert np.all(np.diff(scores[b, :v]) <= 0) and np.all(scores[b, :v] > thr)
        assert np.all(boxes[b, :v, 2] >= boxes[b, :v, 0]) and np.all(boxes[b, :v, 3] >= boxes[b, :v, 1])
        assert not boxes[b, v:].any() and not scores[b, v:].any() and not classes[b, v:].any()
        pairs = list(zip(kept[b, :v].tolist(), classes[b, :v].tolist()))
        assert len(set(pairs)) == v and all(0 <= c < ncls for _, c in pairs)
    for i in (0, 5, 17, 31):
        one = [o.cpu().numpy() for o in eng.predict_device(dev[i:i + 1])]
        for a, b in zip(run1, one):
            assert np.array_equal(a[i:i + 1], b), i
    eng.close()


def test_scheduling_knobs_do_not_change_results():
    """Autotuned tiles, restored tiles and sub-batched early layers are speed knobs: outputs stay bit-identical."""
    cfg, plan, ws, imgs, eng = _setup(160, 3, 3, "bf16", seed=4)
    base = eng.predict(imgs, with_indices=True)
    heads = eng.forward_heads(imgs)
    tiles = eng.autotune(3, reps=1)
    assert len(tiles) == 110 and tiles[0] == 0 and all(t > 0 for t in tiles[1:])
    for a, b in zip(base, eng.predict(imgs, with_indices=True)):
        assert np.array_equal(a, b)
    eng.set_tiles([0] * 110)
    eng.set_subbatch(2, 16)
    for a, b in zip(heads, eng.forward_heads(imgs)):
        assert np.array_equal(a, b)
    for a, b in zip(base, eng.predict(imgs, with_indices=True)):
        assert np.array_equal(a, b)
    eng.set_subbatch(0)
    eng.set_tiles(tiles)
    for a, b in zip(base, eng.predict(imgs, with_indices=True)):
        assert np.array_equal(a, b)
    eng.close()


@pytest.mark.parametrize("dtype", ["f32", "bf16"])
def test_splitk_latency_schedule_vs_oracle(dtype):
    """The latency schedule of `Yolov4.predict(one image)` (reference models.py:109-127): with y4_set_splitk the tuner may split
    the K loop of the deep, few-tile layers over several workgroups.  A split launch sums in another fp32 order, so this schedule
    is NOT in the bit-identical set: it is held to the oracle like any other path -- fp32: heads within 3e-3, detections equal to
    the oracle's within 1e-3 away from the thresholds (the north_star bar); bf16: inside the 16-bit budget -- and to the unsplit
    schedule's own output within a few ulps of fp32 accumulation.  Forced split ids on every layer that accepts them make sure
    the fixup path itself runs even if the tuner of this box prefers unsplit tiles."""
    from oracle import forward as OF, decode_nms as OD
    size, ncls, n = 416, 3, 1
    cfg, plan, ws, imgs, eng = _setup(size, ncls, n, dtype, seed=2)
    base_heads = eng.forward_heads(imgs)
    eng.set_splitk(True)
    tiles = eng.autotune(n, reps=2)
    tuned_split = [t for t in tiles if t >= 100]
    got_heads = eng.forward_heads(imgs)
    res = eng.predict(imgs, with_indices=True)
    want_heads = [np.asarray(h) for h in OF.yolo_model_forward(imgs, ws, ncls)] if dtype == "f32" else None
    for k_, (a, b) in enumerate(zip(got_heads, base_heads)):
        d = np.abs(a - b)
        if dtype == "f32":
            assert d.max() < 1e-3, (k_, d.max())
            assert np.abs(a - want_heads[k_]).max() < 3e-3
        else:
            assert d.mean() < 0.08 and np.quantile(d, 0.999) < 0.45, (k_, d.mean())
    if dtype == "f32":
        rb, rs, rc, rv, ri = OD.inference_from_heads([np.asarray(h) for h in want_heads], ncls, cfg["anchors"], cfg["xyscale"], size)
        boxes, scores, classes, valid, kept = res
        _same_detections(kept[0], classes[0], scores[0], boxes[0], int(valid[0]), ri[0], rc[0], rs[0], rb[0], int(rv[0]),
                         cfg["score_threshold"])
    # forced: a 2-way split of the 64x64 tile wherever the launcher accepts it (K long enough, plain launch)
    ok = 0
    probe = list(tiles)
    for i in range(1, 110):
        probe[i] = 110
        try:
            eng.set_tiles(probe)
            eng.forward_heads(imgs)
            ok += 1
        except Exception:
            probe[i] = tiles[i]
    eng.set_tiles(probe)
    forced_heads = eng.forward_heads(imgs)
    assert ok >= 60, ok
    for a, b in zip(forced_heads, base_heads):
        d = np.abs(a - b)
        assert (d.max() < 1e-3) if dtype == "f32" else (d.mean() < 0.08)
    print(f"split-K {dtype}: tuner chose {len(tuned_split)} split ids {sorted(set(tuned_split))}; forced on {ok} layers")
    eng.close()


@pytest.mark.parametrize("dtype", ["bf16", "f16"])
@pytest.mark.parametrize("size,n", [(96, 5), (160, 3), (416, 2), (608, 1), (640, 1)])
def test_stem_fusion_is_bit_identical(dtype, size, n):
    """convs 0+1 as one kernel (conv 0's output kept in LDS): same MFMA products summed in the same order as the two
    separate kernels, so conv 1's output, the heads and the detections must be bit-identical -- including the
    top/left zero padding of the stride-2 conv, the image border rows and a sub-batched schedule."""
    import yolo4hip.ext as ext
    cfg, plan, ws, imgs, eng = _setup(size, 3, n, dtype, seed=5)
    heads = eng.forward_heads(imgs)
    c1 = eng.conv_output(1, n)
    c0 = eng.conv_output(0, n)
    base = eng.predict(imgs, with_indices=True)
    eng.set_stem_fusion(True)
    for a, b in zip(heads, eng.forward_heads(imgs)):
        assert np.array_equal(a, b)
    assert np.array_equal(c1, eng.conv_output(1, n))
    with pytest.raises(ext.Y4Error):
        eng.conv_output(0, n)
    if n > 1:
        eng.set_subbatch(1, 16)
        for a, b in zip(base, eng.predict(imgs, with_indices=True)):
            assert np.array_equal(a, b)
        eng.set_subbatch(0)
    eng.autotune(n, reps=1)
    for a, b in zip(base, eng.predict(imgs, with_indices=True)):
        assert np.array_equal(a, b)
    eng.set_stem_fusion(False)
    eng.forward_heads(imgs)
    assert np.array_equal(c0, eng.conv_output(0, n))
    eng.close()


@pytest.mark.parametrize("dtype", ["bf16", "f16"])
@pytest.mark.parametrize("size,n", [(352, 2), (608, 1)])
def test_halo_tile_as_lds_pair_head_is_bit_identical(dtype, size, n):
    """The 3x3+Add -> next block's 1x1 runs of the 38^2 stage (convs 42-43 .. 56-57, reference custom_layers.py:34-44) with the 192 x 256
    HALO tile (id 52, csrc/conv_halo_kernel.h) forced as head of the LDS pair: the head's tile stays in LDS over the dead halo buffers and
    the 1x1 conv runs from it.  Heads, the runs' outputs and the detections equal the unfused path bit for bit."""
    cfg, plan, ws, imgs, eng = _setup(size, 3, n, dtype, seed=8)
    heads = eng.forward_heads(imgs)
    taps = (22, 24, 26, 28, 30, 32, 34, 36, 37, 43, 45, 47, 49, 51, 53, 55, 57, 58)
    ref = {i: eng.conv_output(i, n) for i in taps}
    base = eng.predict(imgs, with_indices=True)
    eng.set_chain_fusion(True)
    tiles = [0] * 110
    for head in range(42, 57, 2):
        tiles[head] = -(52 + 1000 * 51)          # run tile 52 (halo pair head); own tile when not fused: the 384 x 128 halo tile
    for head in range(21, 36, 2):                # ... and the 76^2 stage's runs (Cout = 128) with the 384 x 128 / 320 x 128 halo tiles as heads
        tiles[head] = -((51 if size == 352 else 54) + 1000 * 8)
    eng.set_tiles(tiles)
    for a, b in zip(heads, eng.forward_heads(imgs)):
        assert np.array_equal(a, b)
    for i in taps:
        assert np.array_equal(ref[i], eng.conv_output(i, n)), i
    for a, b in zip(base, eng.predict(imgs, with_indices=True)):
        assert np.array_equal(a, b)
    eng.close()


@pytest.mark.parametrize("dtype", ["bf16", "f16"])
@pytest.mark.parametrize("size,n", [(96, 5), (160, 3), (416, 2), (608, 1)])
def test_chain_fusion_is_bit_identical(dtype, size, n):
    """3x3+Add -> 1x1 (-> 1x1 over the concat) and CSP-pair -> 1x1 runs as one kernel each: a chained conv issues the
    same MFMAs on the same 16-bit inputs in the same order as its own kernel would, so every materialised tensor, the
    heads and the detections are bit-identical to the unfused path.  25 runs exist in the plan: three through registers
    (convs 5-6-7, 12-13, 14-15-16) and 22 through an LDS-resident tile (the CSP pair 2+3 -> 4; 3x3+Add -> the next 1x1 in the 76^2 and
    38^2 stages: 21-22 .. 35-36, 42-43 .. 56-57; the neck's 88-89, 90-91 and 92-93 = raw head 0; the stage openers
    8 -> 9+10 and 17 -> 18+19, whose tail is the fused CSP pair); autotune (which may turn a run off again), restored
    tiles and sub-batching keep that."""
    cfg, plan, ws, imgs, eng = _setup(size, 3, n, dtype, seed=6)
    heads = eng.forward_heads(imgs)
    taps = (2, 3, 4, 7, 9, 10, 11, 12, 13, 16, 18, 19, 21, 22, 35, 36, 37, 42, 43, 57, 58, 89, 91, 93, 94)   # outputs the runs still write
    ref = {i: eng.conv_output(i, n) for i in taps}
    base = eng.predict(imgs, with_indices=True)
    assert eng.set_chain_fusion(True) == 26      # 25 runs + the alternative run of the 152^2 stage (conv_chain.h CFG 4)

    def check():
        for a, b in zip(heads, eng.forward_heads(imgs)):
            assert np.array_equal(a, b)
        for i in taps:
            assert np.array_equal(ref[i], eng.conv_output(i, n)), i
        for a, b in zip(base, eng.predict(imgs, with_indices=True)):
            assert np.array_equal(a, b)

    check()
    tiles = eng.autotune(n, reps=1)         # also decides per run: one kernel (reported as -tile) or separate kernels
    heads_of_runs = {0, 2, 5, 8, 12, 14, 15, 17, 88, 90, 92} | set(range(21, 36, 2)) | set(range(42, 57, 2))
    assert all(t != 0 for t in tiles[1:]) and all(t > 0 or i in heads_of_runs for i, t in enumerate(tiles))
    check()
    eng.set_tiles([0] * 110)                # every run chained, built-in tiles
    if n > 1:
        eng.set_subbatch(1, 16)
    check()
    eng.set_subbatch(0)
    eng.set_tiles(tiles)
    check()
    eng.set_chain_fusion(False)
    check()
    eng.close()


@pytest.mark.parametrize("dtype", ["bf16", "f16"])
@pytest.mark.parametrize("size,n", [(96, 5), (160, 3), (416, 2), (608, 2)])
def test_stage_fusion_is_bit_identical(dtype, size, n):
    """Convs 2..7 (the first CSP stage, reference custom_layers.py:47-69 + :105) as ONE spatially tiled kernel
    (csrc/csp_stage.hip): 16x16-pixel tiles with a recomputed 1-pixel halo ring, every intermediate in LDS / registers.
    Each conv issues the same MFMAs on the same 16-bit inputs in the same order as its own kernel, so conv 7's output, every
    later tensor, the heads and the detections must be bit-identical to the unfused path -- including the image border
    (conv 5's zero padding is applied to conv 4's OUTPUT, not to the input tile), tiles at every corner, batches that
    straddle the per-XCD tile ranges, and in combination with the stem fusion, the chain fusions, autotune and
    sub-batching.  While active, convs 2..6 are not materialised."""
    import yolo4hip.ext as ext
    cfg, plan, ws, imgs, eng = _setup(size, 3, n, dtype, seed=8)
    heads = eng.forward_heads(imgs)
    taps = (1, 7, 8, 16, 37)
    ref = {i: eng.conv_output(i, n) for i in taps}
    base = eng.predict(imgs, with_indices=True)

    def check():
        for a, b in zip(heads, eng.forward_heads(imgs)):
            assert np.array_equal(a, b)
        for i in taps:
            assert np.array_equal(ref[i], eng.conv_output(i, n)), i
        for a, b in zip(base, eng.predict(imgs, with_indices=True)):
            assert np.array_equal(a, b)

    assert eng.set_stage_fusion(True) and eng.stage_fusion_active()
    check()
    for idx in (2, 3, 4, 5, 6):
        with pytest.raises(ext.Y4Error):
            eng.conv_output(idx, n)
    eng.set_stem_fusion(True)
    eng.set_chain_fusion(True)
    check()
    if n > 1:
        eng.set_subbatch(1, 16)
        check()
        eng.set_subbatch(0)
    eng.autotune(n, reps=1)                  # keeps the stage kernel only if it measures faster; either way:
    check()
    assert eng.set_stage_fusion(True)        # force it back on with the tuned tiles around it
    check()
    assert not eng.set_stage_fusion(False) and not eng.stage_fusion_active()
    check()
    eng.close()


@pytest.mark.parametrize("dtype", ["bf16", "f16"])
@pytest.mark.parametrize("size,n", [(96, 5), (160, 3), (416, 2), (608, 2)])
def test_res_fusion_is_bit_identical(dtype, size, n):
    """Residual blocks "1x1 -> 3x3 + Add" of the 64- and 128-channel stages (reference custom_layers.py:34-44: convs 11-12,
    13-14 and 20-21 .. 34-35) as one spatially tiled kernel each (csrc/resblock.hip): the 1x1 conv runs in place on the
    LDS-resident halo'd tile, the 3x3 reads its nine taps from it.  Same MFMAs on the same 16-bit inputs in the same K
    order as the separate kernels -> every materialised tensor, the heads and the detections are bit-identical, including
    partially filled edge tiles (sides 12, 20, 24, 40, 52, 76 are not multiples of the 16-pixel tile), the image border
    (zero padding applies to the 1x1 conv's OUTPUT) and the existing chain fusions whose heads / tails the blocks take
    over.  While a block runs fused its 1x1 conv is not materialised."""
    import yolo4hip.ext as ext
    cfg, plan, ws, imgs, eng = _setup(size, 3, n, dtype, seed=9)
    heads = eng.forward_heads(imgs)
    taps = (10, 12, 14, 16, 19, 21, 23, 29, 35, 36, 37)
    ref = {i: eng.conv_output(i, n) for i in taps}
    base = eng.predict(imgs, with_indices=True)

    def check():
        for a, b in zip(heads, eng.forward_heads(imgs)):
            assert np.array_equal(a, b)
        for i in taps:
            assert np.array_equal(ref[i], eng.conv_output(i, n)), i
        for a, b in zip(base, eng.predict(imgs, with_indices=True)):
            assert np.array_equal(a, b)

    assert eng.set_res_fusion(True) == 10 and eng.res_fusion_mask() == 3
    check()
    for idx in (11, 13, 20, 34):
        with pytest.raises(ext.Y4Error):
            eng.conv_output(idx, n)
    eng.set_stem_fusion(True)
    eng.set_chain_fusion(True)
    eng.set_stage_fusion(True)
    check()
    if n > 1:
        eng.set_subbatch(1, 16)
        check()
        eng.set_subbatch(0)
    for mask in (1, 2):                      # one channel group fused, the other on its chains
        eng.set_res_fusion_mask(mask)
        assert eng.res_fusion_mask() == mask
        check()
    eng.set_res_fusion(True)
    eng.autotune(n, reps=1)                  # keeps a group only where it measures faster; either way:
    check()
    eng.set_res_fusion(False)
    assert eng.res_fusion_mask() == 0
    check()
    eng.close()


def test_stem_fusion_rejected_where_unsupported():
    import yolo4hip.ext as ext
    cfg, plan, ws, imgs, eng = _setup(160, 3, 1, "f32")
    with pytest.raises(ext.Y4Error):
        eng.set_stem_fusion(True)
    with pytest.raises(ext.Y4Error):
        eng.set_chain_fusion(True)
    with pytest.raises(ext.Y4Error):
        eng.set_stage_fusion(True)
    with pytest.raises(ext.Y4Error):
        eng.set_res_fusion(True)
    eng.close()


def test_packed_weight_cache_roundtrip(tmp_path):
    from yolo4hip.config import make_config
    from yolo4hip.engine import Engine
    cfg, plan, ws, imgs, eng = _setup(96, 2, 2, "f16", seed=6)
    want = eng.forward_heads(imgs)
    path = str(tmp_path / "packed.bin")
    eng.save_packed(path)
    eng2 = Engine(2, make_config(96), max_batch=2, dtype="f16")
    eng2.load_packed(path)
    for a, b in zip(want, eng2.forward_heads(imgs)):
        assert np.array_equal(a, b)
    eng3 = Engine(3, make_config(96), max_batch=2, dtype="f16")
    with pytest.raises(ValueError):
        eng3.load_packed(path)
    for e in (eng, eng2, eng3):
        e.close()


def test_batch_beyond_the_2gib_descriptor_range_runs_in_chunks():
    """the conv kernels address their input through a 2 GiB raw buffer descriptor.  An fp32 engine at 608x608 with
    46 images has a 2.18 GB conv-0 output: such an op now runs as consecutive image chunks instead of failing with EINVAL on
    the first forward.  Images are independent, so rows 0, 22 and 45 of the big batch must equal those images run alone."""
    import torch
    size, ncls, n = 608, 3, 46
    cfg, plan, ws, imgs, eng = _setup(size, ncls, 3, "f32", seed=2)
    one = [eng.predict(imgs[i:i + 1], with_indices=True) for i in range(3)]
    eng.close()
    from yolo4hip import weights as W
    from yolo4hip.engine import Engine
    big = Engine(ncls, cfg, max_batch=n, dtype="f32")
    big.load_weight_blob(W.flatten(ws))
    batch = torch.zeros((n, size, size, 3), dtype=torch.float32, device=big.device)
    rows = (0, 22, 45)
    for k, r in enumerate(rows):
        batch[r] = torch.from_numpy(imgs[k]).to(big.device)
    outs = [o.cpu().numpy() for o in big.predict_device(batch)]
    for k, r in enumerate(rows):
        for a, b in zip(outs, one[k]):
            assert np.array_equal(a[r:r + 1], b), (r,)
    big.close()


def test_engine_fails_loudly():
    from yolo4hip import ext
    from yolo4hip.config import make_config
    from yolo4hip.engine import Engine
    with pytest.raises(AssertionError):
        Engine(3, make_config(400), max_batch=1)          # not a multiple of 32 (reference models.py:24)
    eng = Engine(3, make_config(96), max_batch=1, dtype="f32")
    import torch
    x = torch.zeros((1, 96, 96, 3), device="cuda")
    with pytest.raises(ext.Y4Error):                        # weights not packed yet
        eng.forward_device(x)
    with pytest.raises(ValueError):
        eng.forward_heads(np.zeros((1, 64, 64, 3), np.float32))
    eng.close()
