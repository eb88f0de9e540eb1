"""The C-ABI library loads without a GPU, exports every symbol include/yolo4hip.h declares, and its
host-only entry points report errors the documented way (no compute calls here)."""
import ctypes as C
import os
import re

from helpers import ROOT


def _header_symbols():
    text = open(os.path.join(ROOT, "include", "yolo4hip.h")).read()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    return sorted(set(re.findall(r"\b(y4_[a-z0-9_]+)\s*\(", text)))


def test_every_declared_symbol_is_exported_and_bound():
    from yolo4hip import ext
    lib = ext.load()
    declared = _header_symbols()
    assert len(declared) >= 25
    for name in declared:
        assert hasattr(lib, name), f"{name} declared in yolo4hip.h but not exported"
        assert name in ext.SYMBOLS, f"{name} has no ctypes prototype in ext.SYMBOLS"
    assert sorted(ext.SYMBOLS) == declared
    assert lib.y4_version().decode().startswith("yolo4hip")
    assert lib.y4_conv_tile_count() >= 4


def test_struct_layouts_match_header_sizes():
    from yolo4hip import ext
    assert C.sizeof(ext.y4_config) == 4 * 4 + 18 * 4 + 3 * 4 + 3 * 4 + 2 * 4 + 2 * 4
    assert C.sizeof(ext.y4_layer_desc) == 9 * 4 + 4 + 8
    assert C.sizeof(ext.y4_conv_desc) == 17 * 4 + 4 + 6 * 8 + 8 + 8 + 3 * 4 + 4 + 8 + 8 + 8   # ... tile+pad, out2, 3 ints, pad, split-K ws, wt_frag


def test_error_reporting_host_only():
    from yolo4hip import ext
    from yolo4hip.config import make_config
    from yolo4hip.engine import _cfg_struct
    lib = ext.load()
    h = C.c_void_p()
    bad = _cfg_struct(make_config(416), 80, 1, "f32")
    bad.img_size = 400                               # reference models.py:24 assert
    assert lib.y4_create(C.byref(bad), C.byref(h)) == -22
    assert b"multiple" in lib.y4_last_error()
    bad = _cfg_struct(make_config(416), 80, 1, "f32")
    bad.num_classes = 0                              # reference models.py:38 assert
    assert lib.y4_create(C.byref(bad), C.byref(h)) == -22
    assert b"no classes detected" in lib.y4_last_error()
    ok = _cfg_struct(make_config(64), 2, 1, "f16")
    assert lib.y4_create(C.byref(ok), C.byref(h)) == 0
    a, w = C.c_size_t(), C.c_size_t()
    assert lib.y4_workspace_bytes(h, C.byref(a), C.byref(w)) == 0 and a.value > 0 and w.value > 0
    assert lib.y4_forward(h, None, 1, None) == -1    # Y4_ESTATE: workspace not bound
    assert b"workspace not bound" in lib.y4_last_error()
    assert lib.y4_bind_workspace(h, None, 0, None, 0) == -22
    assert lib.y4_layer_info(h, 110, None) == -22
    assert lib.y4_destroy(h) == 0


def test_missing_library_fails_loudly(monkeypatch, tmp_path):
    from yolo4hip import ext
    monkeypatch.setattr(ext, "_lib", None)
    monkeypatch.setattr(ext, "LIB_PATH", str(tmp_path / "nope.so"))
    try:
        ext.load()
    except ImportError as e:
        assert "no CPU fallback" in str(e)
    else:
        raise AssertionError("ext.load() must raise when the shared library is missing")


def test_workspace_aliasing_layout_host_only():
    """y4_set_workspace_aliasing is pure host work (liveness over the plan, interval placement): without a GPU it must
    shrink the activation workspace to well under half for every BASELINE shape, leave the weight workspace alone, be
    reversible before the workspace is bound, and refuse sub-batching while on."""
    from yolo4hip import ext
    from yolo4hip.config import make_config
    from yolo4hip.engine import _cfg_struct
    lib = ext.load()
    for size, ncls, nb, dt in ((608, 80, 32, "bf16"), (416, 3, 64, "f16"), (608, 80, 1, "f32"), (416, 80, 2, "f32")):
        cfg = _cfg_struct(make_config(size), ncls, nb, dt)
        h = C.c_void_p()
        ext.check(lib.y4_create(C.byref(cfg), C.byref(h)))
        sizes = []
        for on in (0, 1, 0, 1):
            ext.check(lib.y4_set_workspace_aliasing(h, on))
            a, w = C.c_size_t(), C.c_size_t()
            ext.check(lib.y4_workspace_bytes(h, C.byref(a), C.byref(w)))
            sizes.append((a.value, w.value))
        assert sizes[0] == sizes[2] and sizes[1] == sizes[3]
        assert sizes[1][1] == sizes[0][1]                       # weights untouched
        assert sizes[1][0] < 0.45 * sizes[0][0], (size, nb, dt, sizes)
        assert lib.y4_set_subbatch(h, 1, 16) == -1 and b"aliasing" in lib.y4_last_error()
        ext.check(lib.y4_set_workspace_aliasing(h, 0))
        if nb > 1:
            ext.check(lib.y4_set_subbatch(h, 1, 16))
            assert lib.y4_set_workspace_aliasing(h, 1) == -1    # ... and the other way round
        lib.y4_destroy(h)


def test_every_shipped_schedule_is_accepted_by_the_library():
    """yolo4hip/schedules/*.json (what bench.py and the facade load instead of tuning) against the library's own rules, host
    only: right length, every tile id known, chained heads (negative ids) only where the plan has a chain, stage / residual
    switches settable -- y4_set_tiles and friends validate all of that without a GPU."""
    import glob
    import json
    from yolo4hip import ext
    from yolo4hip.config import make_config
    from yolo4hip.engine import _cfg_struct
    lib = ext.load()
    files = sorted(glob.glob(os.path.join(ROOT, "yolo-v4-tf.keras_amd", "yolo4hip", "schedules", "*.json")))
    assert len(files) >= 7 and {"416_80_32_bf16.json", "608_80_1_bf16.json", "608_80_1_f32.json", "416_80_1_bf16.json",
                                "416_80_1_f32.json"} <= {os.path.basename(f) for f in files}
    for f in files:
        s = json.load(open(f))
        assert os.path.basename(f) == f"{s['size']}_{s['classes']}_{s['batch']}_{s['dtype']}.json"
        cfg = _cfg_struct(make_config(s["size"]), s["classes"], s["batch"], s["dtype"])
        h = C.c_void_p()
        ext.check(lib.y4_create(C.byref(cfg), C.byref(h)))
        f32 = s["dtype"] == "f32"                          # the fp32 (parity) path has tiles only: no fused kernels
        if f32:
            assert not s["stage_fusion"] and s["res_fusion_mask"] == 0 and all(t >= 0 for t in s["tiles"])
        else:
            ext.check(lib.y4_set_stem_fusion(h, 1))
            assert lib.y4_set_chain_fusion(h, 1) > 0
            assert lib.y4_set_stage_fusion(h, 1) == 1
            assert lib.y4_set_res_fusion(h, 1) > 0
        if s.get("splitk"):                                # latency schedules: split-K ids (base + 100 e) on plain launches only
            assert s["batch"] <= 2 and any(t >= 100 for t in s["tiles"]) and all(t < 400 for t in s["tiles"])
        else:
            assert all(abs(t) % 1000 < 100 for t in s["tiles"])
        tiles = (C.c_int32 * len(s["tiles"]))(*s["tiles"])
        assert len(s["tiles"]) == lib.y4_num_layers(h) == 110
        ext.check(lib.y4_set_tiles(h, tiles, len(s["tiles"])))
        back = (C.c_int32 * 110)()
        ext.check(lib.y4_get_tiles(h, back, 110))
        # the handle reports exactly the file, and what it reports restores it exactly (a run's head carries its run tile AND its
        # own tile in one entry): set(get()) is the identity, also on a fresh handle with the run switched off before
        assert list(back) == [int(t) for t in s["tiles"]], [(i, a, b) for i, (a, b) in enumerate(zip(s["tiles"], back)) if a != b]
        h2 = C.c_void_p()
        ext.check(lib.y4_create(C.byref(cfg), C.byref(h2)))
        ext.check(lib.y4_copy_schedule(h, h2))
        back2 = (C.c_int32 * 110)()
        ext.check(lib.y4_get_tiles(h2, back2, 110))
        assert list(back2) == list(back) and lib.y4_get_stage_fusion(h2) == lib.y4_get_stage_fusion(h) \
            and lib.y4_get_res_fusion(h2) == lib.y4_get_res_fusion(h)
        # a run head's own tile survives the round trip: give conv 15 (the alternative run's head) one, with the run in force
        if back[15] < 0:
            probe = list(back)
            probe[15] = -((-back[15]) % 1000 + 1000 * 7)
            arr = (C.c_int32 * 110)(*probe)
            ext.check(lib.y4_set_tiles(h, arr, 110))
            ext.check(lib.y4_get_tiles(h, back2, 110))
            assert list(back2) == probe
            ext.check(lib.y4_set_tiles(h2, back2, 110))
            back3 = (C.c_int32 * 110)()
            ext.check(lib.y4_get_tiles(h2, back3, 110))
            assert list(back3) == probe
            ext.check(lib.y4_set_tiles(h, tiles, len(s["tiles"])))       # a plain -t (older files) leaves the own tile: still 7
            ext.check(lib.y4_get_tiles(h, back2, 110))
            own = (-s["tiles"][15]) // 1000 or 7
            assert back2[15] == -((-s["tiles"][15]) % 1000 + 1000 * own)
        assert lib.y4_copy_schedule(h, h) < 0
        lib.y4_destroy(h2)
        if not f32:
            assert lib.y4_set_stage_fusion(h, int(s["stage_fusion"])) == int(s["stage_fusion"])
            ext.check(lib.y4_set_res_fusion_mask(h, int(s["res_fusion_mask"])))
            assert lib.y4_get_res_fusion(h) == s["res_fusion_mask"]
        lib.y4_destroy(h)
