"""Host logic: the 110-conv plan (yolo4hip/plan.py <- reference custom_layers.py:100-198) against the
numbers of SURVEY.md Appendix A, and against the C++ plan through the C ABI (host-only calls, no GPU)."""
import ctypes as C

import pytest

from yolo4hip.plan import ACT_LEAKY, ACT_LINEAR, ACT_MISH, build_plan


def test_totals_match_survey_appendix_a():
    p = build_plan(608, 80)
    assert len(p.convs) == 110
    assert abs(p.flops_per_image / 1e9 - 128.389) < 1e-3
    assert p.n_params == 64429405                      # (257717640 - 20) / 4: the published yolov4.weights size
    assert p.num_boxes == 22743 and p.grids == (76, 38, 19)
    assert abs(build_plan(416, 3).flops_per_image / 1e9 - 59.545) < 1e-3
    assert abs(build_plan(416, 80).flops_per_image / 1e9 - 60.105) < 1e-3
    assert abs(build_plan(608, 3).flops_per_image / 1e9 - 127.194) < 1e-3
    assert build_plan(416, 80).num_boxes == 10647
    k1 = sum(1 for c in p.convs if c.k == 1)
    s1 = sum(1 for c in p.convs if c.k == 3 and c.s == 1)
    s2 = sum(1 for c in p.convs if c.k == 3 and c.s == 2)
    assert (k1, s1, s2) == (66, 37, 7)
    acts = [c.act for c in p.convs]
    assert (acts.count(ACT_MISH), acts.count(ACT_LEAKY), acts.count(ACT_LINEAR)) == (70, 37, 3)
    assert sum(1 for c in p.convs if c.bn) == 107
    assert sum(1 for o in p.ops if o.kind == "add") == 23
    assert sum(1 for o in p.ops if o.kind == "concat") == 10
    assert sum(1 for o in p.ops if o.kind == "maxpool") == 3
    assert sum(1 for o in p.ops if o.kind == "upsample") == 2
    backbone = sum(c.flops_per_image for c in p.convs[:72]) / 1e9
    assert abs(backbone - 73.696) < 1e-3


@pytest.mark.parametrize("idx,k,s,cin,cout,act,bn,out_side", [
    (0, 3, 1, 3, 32, ACT_LEAKY, True, 608),        # stem is LEAKY in the reference (custom_layers.py:101)
    (1, 3, 2, 32, 64, ACT_LEAKY, True, 304),
    (2, 1, 1, 64, 64, ACT_MISH, True, 304),        # csp1 route conv is created first (:59)
    (4, 1, 1, 64, 32, ACT_MISH, True, 304),        # bottleneck halves the 1x1 width (:63)
    (5, 3, 1, 32, 64, ACT_MISH, True, 304),
    (7, 1, 1, 128, 64, ACT_MISH, True, 304),
    (8, 3, 2, 64, 128, ACT_MISH, True, 152),
    (37, 1, 1, 256, 256, ACT_MISH, True, 76),
    (58, 1, 1, 512, 512, ACT_MISH, True, 38),
    (71, 1, 1, 1024, 1024, ACT_MISH, True, 19),
    (72, 1, 1, 1024, 512, ACT_LEAKY, True, 19),
    (75, 1, 1, 2048, 512, ACT_LEAKY, True, 19),    # consumes the SPP concat
    (78, 1, 1, 512, 256, ACT_LEAKY, True, 19),
    (79, 1, 1, 512, 256, ACT_LEAKY, True, 38),     # lateral conv created after the upsample-branch conv
    (93, 1, 1, 256, 255, ACT_LINEAR, False, 76),
    (94, 3, 2, 128, 256, ACT_LEAKY, True, 38),
    (101, 1, 1, 512, 255, ACT_LINEAR, False, 38),
    (102, 3, 2, 256, 512, ACT_LEAKY, True, 19),
    (109, 1, 1, 1024, 255, ACT_LINEAR, False, 19),
])
def test_rows_of_appendix_a(idx, k, s, cin, cout, act, bn, out_side):
    c = build_plan(608, 80).convs[idx]
    assert (c.k, c.s, c.cin, c.cout, c.act, c.bn, c.out_side) == (k, s, cin, cout, act, bn, out_side)


def test_concat_orders():
    p = build_plan(416, 80)
    cats = [o for o in p.ops if o.kind == "concat"]
    assert cats[0].srcs == ("c6", "c2")                                    # CSP: [x, route]
    spp = cats[5]
    assert [p.ops[[o.dst for o in p.ops].index(s)].k for s in spp.srcs[:3]] == [13, 9, 5] and spp.srcs[3] == "c74"
    assert cats[6].srcs[0] == "c79" and cats[6].srcs[1].startswith("up")   # [lateral, upsampled]
    assert cats[8].srcs == ("c94", "c84")                                  # [downsampled, route1']
    assert cats[9].srcs == ("c102", "c77")                                 # [downsampled, route_input]
    assert p.chans[spp.dst] == 2048


def test_reference_asserts():
    with pytest.raises(AssertionError):
        build_plan(400, 80)
    with pytest.raises(AssertionError):
        build_plan(416, 0)


@pytest.mark.parametrize("size,ncls", [(416, 80), (608, 3), (96, 6)])
def test_cpp_plan_equals_python_plan_host_only(size, ncls):
    """y4_create / y4_layer_info are host-only, so the C++ plan can be checked on a machine without a GPU."""
    from yolo4hip import ext
    from yolo4hip.config import make_config
    from yolo4hip.engine import _cfg_struct
    lib = ext.load()
    cfg = _cfg_struct(make_config(size), ncls, 4, "bf16")
    h = C.c_void_p()
    assert lib.y4_create(C.byref(cfg), C.byref(h)) == 0
    plan = build_plan(size, ncls)
    assert lib.y4_num_layers(h) == 110
    off = 0
    for c in plan.convs:
        d = ext.y4_layer_desc()
        assert lib.y4_layer_info(h, c.idx, C.byref(d)) == 0
        assert (d.ksize, d.stride, d.cin, d.cout, d.act, d.has_bn, d.in_side, d.out_side, d.weight_offset) == \
               (c.k, c.s, c.cin, c.cout, c.act, int(c.bn), c.in_side, c.out_side, off)
        off += (4 if c.bn else 1) * c.cout + c.n_weights
    fl, nb, hcs, wf = C.c_int64(), C.c_int32(), C.c_int32(), C.c_int64()
    assert lib.y4_model_info(h, C.byref(fl), C.byref(nb), C.byref(hcs), C.byref(wf)) == 0
    assert (fl.value, nb.value, wf.value) == (plan.flops_per_image, plan.num_boxes, plan.n_params)
    assert hcs.value == (3 * (ncls + 5) + 7) // 8 * 8
    assert lib.y4_destroy(h) == 0
