"""GPU parity at BASELINE.json's real configurations against the oracle, and of the device-side BatchNorm fold.

  * config 3 (608x608, 80 classes, bf16, fusions + autotune on): head logits and detections of 4 images of the batch vs
    the fp32 oracle, with the error budget STATED here (and recorded in LABNOTES.md section 2);
  * config 5 (416x416, 3 classes, fp16) at its real batch of 64: size-independent properties over the whole batch and
    4 images vs the oracle;
  * `y4_pack_weights` -> fold_bn_kernel (csrc/misc_kernels.hip) with random BN mean / var / gamma against the oracle
    (reference utils.py:28-31 row order [beta, gamma, mean, var]; custom_layers.py:26 eps 1e-3).
The measured numbers of each run are appended to gpurun_out/parity_measured.jsonl (scratch) for DESIGN.md.
"""
import json
import os

import numpy as np
import pytest

from helpers import ROOT, detection_agreement, randomize_bn, score_delta_quantile

pytestmark = pytest.mark.gpu


def _record(tag, payload):
    try:
        os.makedirs(os.path.join(ROOT, "gpurun_out"), exist_ok=True)
        with open(os.path.join(ROOT, "gpurun_out", "parity_measured.jsonl"), "a") as f:
            f.write(json.dumps({"test": tag, **payload}) + "\n")
    except OSError:
        pass


def _engine(size, ncls, n, dtype, ws):
    from yolo4hip import weights as W
    from yolo4hip.config import make_config
    from yolo4hip.engine import Engine
    cfg = make_config(size)
    eng = Engine(ncls, cfg, max_batch=n, dtype=dtype)
    eng.load_weight_blob(W.flatten(ws))
    return cfg, eng


def _head_errors(heads, ref_heads, rows):
    out = []
    for a, b in zip(heads, ref_heads):
        err = np.abs(a[rows] - b)
        assert np.isfinite(a).all()
        out.append((float(err.mean()), float(np.quantile(err, 0.999)), float(err.max())))
    return out


# Error budget of 16-bit STORAGE through 110 layers (fp32 accumulate, fp32 BN/activation; every activation tensor rounded
# to 8 (bf16) / 11 (fp16) mantissa bits), on raw head logits of std ~1.3.  Measured on MI355X over rounds 2-5 (every run appends to
# gpurun_out/parity_measured.jsonl; the last ones are kept under profiles/):
#   bf16, 608/80 batch 32: mean |err| 0.062-0.075, 99.9 % quantile 0.27-0.34, max 0.52-0.59; 89-94 % of the oracle's 100
#         detections per image matched by (box, class); score delta of the matched ones: 90 % below 0.024-0.034, max 0.04-0.10
#   fp16, 416/3 batch 64: mean 0.0063-0.0075, 99.9 % quantile 0.033-0.042; 98-100 % matched, score delta <= 0.007 (90 % below 0.0035)
#   fp16, 608/80 batch 32: mean 0.0078-0.0095, 99.9 % quantile 0.035-0.043, max 0.069; 97-100 % matched, score delta <= 0.008
#         (90 % below 0.0035), box delta <= 0.0013 -- eight to ten times closer to the oracle than bf16 at the same MFMA rate
# Bounds (round 5, VERDICT r4 item 4: measured + ~15 % instead of + 40 %):
#   * the BULK figures -- mean, 99.9 % quantile, matched fraction (3 detections of 100 below the worst measured image), the 90 %
#     quantile of the score delta -- are tight;
#   * the TAIL figure, the largest score delta of any matched detection, has a FIXED bound since round 6 (a bound derived
#     from the same run's worst logit error was three times what any run measured): 0.12 for bf16 against 0.058 (round 4), 0.10
#     (round 5, after the K order changed) and 0.10 (round 6, halo2 schedule); 0.010 for fp16 against 0.008.  What the largest logit
#     error allows -- score = sigmoid(obj) sigmoid(cls), d score <= (|d obj| + |d cls|) / 4 <= max logit error / 2 -- stays as a
#     second, looser sanity check.
# Identical kept indices are NOT claimed at 16-bit precision (they are, bit for bit, for the decode/NMS kernels fed the
# oracle's heads, and within near-ties for the fp32 path: tests/test_gpu_forward.py).
BUDGET = {
    #        mean |err|, 99.9 % quantile, min matched fraction, max score delta (tail), 90 % quantile of the score delta (bulk)
    "bf16": (0.087, 0.40, 0.86, 0.12, 0.040),        # (matched: 0.88-0.94 measured over rounds 5-6, shipped halo2 schedule included: two detections of margin)
    "f16": (0.011, 0.050, 0.96, 0.010, 0.0042),
}


def _check_budget(dtype, errs, agree, q90, rows):
    mean_b, q_b, frac_b, ds_b, q90_b = BUDGET[dtype]
    worst_logit = max(e[2] for e in errs)
    for i, (m, q, _) in enumerate(errs):
        assert m < mean_b and q < q_b, f"head {i}: mean {m:.4f} q99.9 {q:.4f}"
    for j, (frac, ds, db) in enumerate(agree):
        assert frac >= frac_b, f"image {rows[j]}: matched {frac:.3f} < {frac_b}"
        assert q90[j] < q90_b, f"image {rows[j]}: 90 % of the matched detections' score deltas should lie below {q90_b}, got {q90[j]:.4f}"
        assert ds < min(ds_b, 0.5 * worst_logit + 1e-3), f"image {rows[j]}: score delta {ds:.4f} exceeds what the largest logit error {worst_logit:.3f} allows"


_ORACLE_CACHE = {}


def _headline_oracle(size, ncls, rows, ws, imgs, cfg):
    """The fp32 oracle's heads and detections for the headline batch's sample images (computed once per process: ~10 s of CPU)."""
    from oracle import forward as OF, decode_nms as OD
    key = (size, ncls, tuple(rows))
    if key not in _ORACLE_CACHE:
        ref_heads = OF.yolo_model_forward(imgs[rows], ws, ncls)
        _ORACLE_CACHE[key] = (ref_heads, OD.inference_from_heads(ref_heads, ncls, cfg["anchors"], cfg["xyscale"], size))
    return _ORACLE_CACHE[key]


@pytest.mark.parametrize("dtype", ["bf16", "f16"])
def test_headline_config_vs_oracle(dtype):
    """BASELINE.json config 3: 608x608, 80 classes, batch 32, stem + chain + stage fusions and autotune on -- the schedule family
    bench.py times -- against oracle.forward (fp32) on 4 images of the batch (indices 0, 9, 18, 31).  bf16 is what BASELINE names;
    fp16 is the same kernels at the same MFMA rate with three more mantissa bits (measured side by side)."""
    import torch
    from yolo4hip import weights as W
    from yolo4hip.plan import build_plan
    size, ncls, n = 608, 80, 32
    ws = W.synth_weights(build_plan(size, ncls), seed=0)
    imgs = W.synth_images(n, size, seed=0)
    cfg, eng = _engine(size, ncls, n, dtype, ws)
    eng.set_stem_fusion(True)
    eng.set_chain_fusion(True)
    eng.set_stage_fusion(True)
    dev = torch.from_numpy(imgs).to(eng.device)
    eng.predict_device(dev)
    eng.autotune(n, reps=2)
    outs = [o.cpu().numpy() for o in eng.predict_device(dev)]
    heads = [h.cpu().numpy() for h in eng.heads_device(n)]
    assert all(np.isfinite(h).all() for h in heads), "non-finite head logits (fp16 overflow?)"
    rows = [0, 9, 18, 31]
    ref_heads, (rb, rs, rc, rv, ri) = _headline_oracle(size, ncls, rows, ws, imgs, cfg)
    errs = _head_errors(heads, ref_heads, rows)
    boxes, scores, classes, valid, kept = outs
    agree = [detection_agreement(kept[r], classes[r], scores[r], boxes[r], valid[r], ri[j], rc[j], rs[j], rb[j], rv[j])
             for j, r in enumerate(rows)]
    q90 = [score_delta_quantile(kept[r], classes[r], scores[r], valid[r], ri[j], rc[j], rs[j], rv[j]) for j, r in enumerate(rows)]
    _record(f"headline_608_80_{dtype}_b32", {"head_err_mean_q999_max": errs, "agreement_frac_dscore_dbox": agree, "dscore_q90": q90,
                                             "valid": [int(valid[r]) for r in rows], "ref_valid": [int(v) for v in rv]})
    _check_budget(dtype, errs, agree, q90, rows)
    assert sum(int(v) for v in rv) > 40, "the synthetic heads must give NMS real work"
    # ... and the schedule that SHIPS for this shape (what the default bench.py and the facade run): since round 6 it holds halo2 tile
    # ids (conv_halo2_kernel.h: k-steps of 16, another fixed fp32 summation order) -- the same budget against the same oracle
    sched = eng.shipped_schedule()
    assert sched is not None and sched.get("halo2") == any(55 <= t <= 62 for t in sched["tiles"])
    eng.set_res_fusion(True)
    eng.apply_schedule(sched)
    outs = [o.cpu().numpy() for o in eng.predict_device(dev)]
    heads = [h.cpu().numpy() for h in eng.heads_device(n)]
    assert all(np.isfinite(h).all() for h in heads)
    errs = _head_errors(heads, ref_heads, rows)
    boxes, scores, classes, valid, kept = outs
    agree = [detection_agreement(kept[r], classes[r], scores[r], boxes[r], valid[r], ri[j], rc[j], rs[j], rb[j], rv[j])
             for j, r in enumerate(rows)]
    q90 = [score_delta_quantile(kept[r], classes[r], scores[r], valid[r], ri[j], rc[j], rs[j], rv[j]) for j, r in enumerate(rows)]
    _record(f"headline_608_80_{dtype}_b32_shipped_schedule", {"halo2": bool(sched.get("halo2")), "head_err_mean_q999_max": errs,
                                                              "agreement_frac_dscore_dbox": agree, "dscore_q90": q90})
    _check_budget(dtype, errs, agree, q90, rows)
    eng.close()


def test_config5_416_b64_f16_real_batch():
    """BASELINE.json config 5 at its real size: 416x416, 3 classes (bccd), fp16, batch 64.  Whole batch: the fused +
    autotuned schedule equals the plain one bit for bit, the step is deterministic, image i of the batch equals image i
    alone; 4 images vs the fp32 oracle within the fp16 budget."""
    import torch
    from oracle import forward as OF, decode_nms as OD
    from yolo4hip import weights as W
    from yolo4hip.plan import build_plan
    size, ncls, n, dtype = 416, 3, 64, "f16"
    ws = W.synth_weights(build_plan(size, ncls), seed=0)
    imgs = W.synth_images(n, size, seed=11)
    cfg, eng = _engine(size, ncls, n, dtype, ws)
    dev = torch.from_numpy(imgs).to(eng.device)
    plain = [o.cpu().numpy() for o in eng.predict_device(dev)]
    eng.set_stem_fusion(True)
    eng.set_chain_fusion(True)
    eng.set_stage_fusion(True)
    eng.autotune(n, reps=1)
    run1 = [o.cpu().numpy() for o in eng.predict_device(dev)]
    heads = [h.cpu().numpy() for h in eng.heads_device(n)]
    run2 = [o.cpu().numpy() for o in eng.predict_device(dev)]
    for a, b, c in zip(run1, run2, plain):
        assert np.array_equal(a, b) and np.array_equal(a, c)
    boxes, scores, classes, valid, kept = run1
    assert valid.sum() > 64 and valid.max() <= 100
    for i in (0, 21, 63):
        one = [o.cpu().numpy() for o in eng.predict_device(dev[i:i + 1])]
        for a, b in zip(run1, one):
            assert np.array_equal(a[i:i + 1], b), i
    rows = [0, 13, 40, 63]
    ref_heads = OF.yolo_model_forward(imgs[rows], ws, ncls)
    errs = _head_errors(heads, ref_heads, rows)
    rb, rs, rc, rv, ri = OD.inference_from_heads(ref_heads, ncls, cfg["anchors"], cfg["xyscale"], size)
    agree = [detection_agreement(kept[r], classes[r], scores[r], boxes[r], valid[r], ri[j], rc[j], rs[j], rb[j], rv[j])
             for j, r in enumerate(rows)]
    q90 = [score_delta_quantile(kept[r], classes[r], scores[r], valid[r], ri[j], rc[j], rs[j], rv[j]) for j, r in enumerate(rows)]
    _record("config5_416_3_f16_b64", {"head_err_mean_q999_max": errs, "agreement_frac_dscore_dbox": agree, "dscore_q90": q90,
                                      "valid": [int(valid[r]) for r in rows], "ref_valid": [int(v) for v in rv]})
    _check_budget(dtype, errs, agree, q90, rows)
    # the schedule that ships for this shape (halo2 ids since round 6): same oracle, same budget, same bits twice
    sched = eng.shipped_schedule()
    assert sched is not None
    eng.set_res_fusion(True)
    eng.apply_schedule(sched)
    run3 = [o.cpu().numpy() for o in eng.predict_device(dev)]
    heads = [h.cpu().numpy() for h in eng.heads_device(n)]
    run4 = [o.cpu().numpy() for o in eng.predict_device(dev)]
    for a, b in zip(run3, run4):
        assert np.array_equal(a, b)
    errs = _head_errors(heads, ref_heads, rows)
    boxes, scores, classes, valid, kept = run3
    agree = [detection_agreement(kept[r], classes[r], scores[r], boxes[r], valid[r], ri[j], rc[j], rs[j], rb[j], rv[j])
             for j, r in enumerate(rows)]
    q90 = [score_delta_quantile(kept[r], classes[r], scores[r], valid[r], ri[j], rc[j], rs[j], rv[j]) for j, r in enumerate(rows)]
    _record("config5_416_3_f16_b64_shipped_schedule", {"halo2": bool(sched.get("halo2")), "head_err_mean_q999_max": errs,
                                                       "agreement_frac_dscore_dbox": agree, "dscore_q90": q90})
    _check_budget(dtype, errs, agree, q90, rows)
    eng.close()


@pytest.mark.parametrize("dtype,size,ncls,n", [("bf16", 416, 3, 2), ("f16", 416, 3, 2), ("bf16", 608, 80, 1)])
def test_16bit_error_is_the_storage_floor(dtype, size, ncls, n):
    """Is the 16-bit error budget above (bf16: mean |err| 0.06-0.075 on the logits) inherent to 16-bit STORAGE or a kernel
    artefact?  The oracle can emulate the storage pipeline on the CPU (oracle.forward `storage=`: weights and every stored
    activation rounded to 16 bits, everything else float32).  Measured and asserted:
      floor   = |emulation - fp32 oracle|   = the whole budget: what rounding 110 layers of activations costs on ANY hardware
                                            (measured bf16 416/3: 0.050 / 0.057 / 0.060 per head; 608/80: 0.063 / 0.075 / 0.076;
                                            fp16 416/3: 0.0063 / 0.0070 / 0.0075)
      total   = |HIP - fp32 oracle|         the same numbers (0.051 / 0.057 / 0.061; 0.063 / 0.076 / 0.077; 0.0065 / 0.0073 / 0.0074)
      kernels = |HIP - emulation|           also the same size (0.046 / 0.053 / 0.055): two 16-bit evaluations that differ in
                                            fp32 summation order (MFMA tiles vs oneDNN) flip 1-ulp roundings, which the depth
                                            amplifies like every other rounding -- each is as far from the other as from fp32
    The HIP path must be no further from the fp32 oracle than the emulation is (factor 1.25): it sits ON the storage floor."""
    from oracle import forward as OF
    from yolo4hip import weights as W
    from yolo4hip.plan import build_plan
    ws = W.synth_weights(build_plan(size, ncls), seed=0)
    imgs = W.synth_images(n, size, seed=0)
    cfg, eng = _engine(size, ncls, n, dtype, ws)
    eng.set_stem_fusion(True)
    eng.set_chain_fusion(True)
    heads = eng.forward_heads(imgs)
    ref = OF.yolo_model_forward(imgs, ws, ncls)
    emu = OF.yolo_model_forward(imgs, ws, ncls, storage=dtype)
    rec = []
    for i in range(3):
        floor = float(np.abs(emu[i] - ref[i]).mean())
        total = float(np.abs(heads[i] - ref[i]).mean())
        kern = float(np.abs(heads[i] - emu[i]).mean())
        rec.append((floor, total, kern))
        assert total <= 1.25 * floor + 1e-4, f"head {i}: HIP {total:.4f} from the fp32 oracle, the storage emulation only {floor:.4f}"
        assert kern <= 1.1 * floor, f"head {i}: HIP is {kern:.4f} from the storage emulation (floor {floor:.4f})"
    _record(f"storage_floor_{size}_{ncls}_{dtype}", {"mean_abs_err_floor_total_kernels_per_head": rec})
    eng.close()


@pytest.mark.parametrize("dtype", ["f32", "bf16"])
def test_bn_fold_with_real_statistics(dtype):
    """fold_bn_kernel through y4_pack_weights with random BN mean (+-0.5), var (0.3..3), gamma (0.5..1.5) on all 107 BN
    layers (helpers.randomize_bn: function-preserving, so the net stays well conditioned) against the oracle, which
    applies gamma*rsqrt(var+eps), beta-mean*scale itself.  Power check: the oracle with `mean` negated is far away."""
    import torch
    from oracle import forward as OF
    from yolo4hip import weights as W
    from yolo4hip.plan import build_plan
    size, ncls, n = 160, 3, 2
    plan = build_plan(size, ncls)
    ws = randomize_bn(W.synth_weights(plan, seed=2), seed=2)
    assert max(float(np.abs(cw.bn[2]).max()) for cw in ws if cw.bn is not None) > 0.45
    imgs = W.synth_images(n, size, seed=2)
    cfg, eng = _engine(size, ncls, n, dtype, ws)
    heads = eng.forward_heads(imgs)
    ref = OF.yolo_model_forward(imgs, ws, ncls)
    if dtype == "f32":
        ref64 = OF.yolo_model_forward(imgs, ws, ncls, dtype=torch.float64)
        for i, (a, b, c) in enumerate(zip(heads, ref, ref64)):
            e_gpu, e_cpu = np.abs(a - c).max(), np.abs(b - c).max()
            assert e_gpu <= 2 * e_cpu + 1e-5, f"head {i}: HIP err vs fp64 {e_gpu:.3e}, oracle fp32 err {e_cpu:.3e}"
            assert np.abs(a - b).max() < 3e-3, f"head {i}: {np.abs(a - b).max():.3e}"
    else:
        for i, (m, q, _) in enumerate(_head_errors(heads, ref, slice(None))):
            assert m < BUDGET["bf16"][0] and q < BUDGET["bf16"][1], f"head {i}: mean {m:.4f} q99.9 {q:.4f}"
    flipped = []
    for cw in ws:
        if cw.bn is None:
            flipped.append(cw)
        else:
            bn = cw.bn.copy(); bn[2] = -bn[2]
            flipped.append(type(cw)(w=cw.w, bn=bn))
    wrong = OF.yolo_model_forward(imgs, flipped, ncls)
    assert max(float(np.abs(a - b).max()) for a, b in zip(heads, wrong)) > 0.5, "test has no power against a mean sign slip"
    eng.close()


def test_shipped_schedule_keeps_the_bits_at_the_headline_shape():
    """608x608 / 80 classes / batch 32 / bf16 with the schedule that ships for this shape (tuned tiles, stage kernel,
    residual-block kernels: what the default bench.py and the Yolov4 facade run) against the built-in heuristic with every
    fusion off, on the liveness-aliased workspace.  Since round 6 the shipped schedule holds halo2 tile ids (conv_halo2_kernel.h:
    v_mfma_32x32x16, k-steps of 16 -- another fixed fp32 summation order; `"halo2": true` in the file), so:
      * the schedule with its halo2 ids put back on the 16x16x32 tiles is BIT-IDENTICAL to the plain path (every other scheduling
        choice still is a pure re-arrangement of the same MFMAs);
      * the schedule as shipped gives the same bits twice, and its heads stay within the distance two fp32 summation orders of
        the same 16-bit pipeline have from each other (the `kernels` figure of test_16bit_error_is_the_storage_floor: of the
        size of the storage floor itself) -- its distance to the ORACLE is held by test_headline_config_vs_oracle."""
    import torch
    from yolo4hip import weights as W
    from yolo4hip.config import make_config
    from yolo4hip.engine import Engine
    from yolo4hip.plan import build_plan
    size, ncls, n = 608, 80, 32
    eng = Engine(ncls, make_config(size), max_batch=n, dtype="bf16", alias_workspace=True)
    sched = eng.shipped_schedule()
    assert sched is not None and len(sched["tiles"]) == 110
    eng.load_weight_blob(W.flatten(W.synth_weights(build_plan(size, ncls), 0)))
    imgs = torch.from_numpy(W.synth_images(n, size, seed=11)).to(eng.device)
    plain = [o.cpu().numpy() for o in eng.predict_device(imgs)]
    plain_heads = [h.cpu().numpy() for h in eng.heads_device(n)]
    eng.set_stem_fusion(True); eng.set_chain_fusion(True); eng.set_stage_fusion(True); eng.set_res_fusion(True)
    is_h2 = lambda t: 55 <= t <= 62
    assert bool(sched.get("halo2")) == any(is_h2(t) for t in sched["tiles"])
    no_h2 = dict(sched)
    no_h2["tiles"] = [51 if is_h2(t) else t for t in sched["tiles"]]          # the 16x16x32 halo tile every one of these layers also fits
    eng.apply_schedule(no_h2)
    assert eng.stage_fusion_active() == bool(sched["stage_fusion"]) and eng.res_fusion_mask() == sched["res_fusion_mask"]
    got = [o.cpu().numpy() for o in eng.predict_device(imgs)]
    got_heads = [h.cpu().numpy() for h in eng.heads_device(n)]
    for a, b in zip(got + got_heads, plain + plain_heads):
        assert np.array_equal(a, b)
    assert int(got[3].sum()) > 0            # there are detections to compare
    eng.apply_schedule(sched)
    first = [h.cpu().numpy() for h in (eng.predict_device(imgs), eng.heads_device(n))[1]]
    again = [h.cpu().numpy() for h in (eng.predict_device(imgs), eng.heads_device(n))[1]]
    for a, b in zip(first, again):
        assert np.array_equal(a, b), "the shipped schedule is not repeatable"
    if sched.get("halo2"):
        for i, (a, b) in enumerate(zip(first, plain_heads)):
            d = np.abs(a - b)
            assert float(d.mean()) < BUDGET["bf16"][0] and float(np.quantile(d, 0.999)) < BUDGET["bf16"][1], \
                f"head {i}: the halo2 schedule is {d.mean():.4f} (mean) / {np.quantile(d, 0.999):.3f} (99.9 %) from the 16x16x32 schedule"
            assert float(d.mean()) > 0, "the halo2 ids changed nothing: is the schedule applied?"
        _record("halo2_vs_16x16x32_schedule_608_80_32_bf16", {"mean_abs_head_delta": [float(np.abs(a - b).mean()) for a, b in zip(first, plain_heads)]})
    eng.close()
