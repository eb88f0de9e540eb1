"""ctypes binding of libyolo4hip.so (the C ABI declared in include/yolo4hip.h).

There is deliberately NO fallback: if the shared library is missing, or a call fails, this raises.
PyTorch is used only to own device memory (`torch.empty(..., device='cuda')`) and streams; every
pointer handed to the library is a `tensor.data_ptr()`.
"""
import ctypes as C
import os

_HERE = os.path.dirname(os.path.abspath(__file__))
# YOLO4HIP_LIB lets kernel experiments (scripts/, scratch builds) load an alternative build of the SAME ABI
LIB_PATH = os.environ.get("YOLO4HIP_LIB") or os.path.join(_HERE, "libyolo4hip.so")

F32, BF16, F16 = 0, 1, 2
DTYPE_IDS = {"f32": F32, "fp32": F32, "float32": F32, "bf16": BF16, "bfloat16": BF16, "f16": F16, "fp16": F16,
             "float16": F16}
DTYPE_NAMES = {F32: "f32", BF16: "bf16", F16: "f16"}
ACT_LINEAR, ACT_LEAKY, ACT_MISH = 0, 1, 2


class Y4Error(RuntimeError):
    def __init__(self, code, msg):
        super().__init__(f"libyolo4hip error {code}: {msg}")
        self.code = code


class y4_config(C.Structure):
    _fields_ = [("img_size", C.c_int32), ("num_classes", C.c_int32), ("max_batch", C.c_int32),
                ("dtype", C.c_int32), ("anchors", C.c_float * 18), ("xyscale", C.c_float * 3),
                ("strides", C.c_int32 * 3), ("iou_threshold", C.c_float), ("score_threshold", C.c_float),
                ("max_per_class", C.c_int32), ("max_total", C.c_int32)]


class y4_layer_desc(C.Structure):
    _fields_ = [("idx", C.c_int32), ("ksize", C.c_int32), ("stride", C.c_int32), ("cin", C.c_int32),
                ("cout", C.c_int32), ("act", C.c_int32), ("has_bn", C.c_int32), ("in_side", C.c_int32),
                ("out_side", C.c_int32), ("weight_offset", C.c_int64)]


class y4_conv_desc(C.Structure):
    _fields_ = [("dtype", C.c_int32), ("n", C.c_int32), ("h", C.c_int32), ("w", C.c_int32), ("cin", C.c_int32),
                ("cout", C.c_int32), ("ksize", C.c_int32), ("stride", C.c_int32), ("act", C.c_int32),
                ("upsample", C.c_int32), ("out_f32", C.c_int32), ("in_cstride", C.c_int32),
                ("in_coff", C.c_int32), ("out_cstride", C.c_int32), ("out_coff", C.c_int32),
                ("res_cstride", C.c_int32), ("res_coff", C.c_int32), ("in_", C.c_void_p), ("wt", C.c_void_p),
                ("scale", C.c_void_p), ("shift", C.c_void_p), ("res", C.c_void_p), ("out", C.c_void_p),
                ("tile", C.c_int32), ("out2", C.c_void_p), ("out2_cstride", C.c_int32), ("out2_coff", C.c_int32),
                ("split", C.c_int32), ("splitk_ws", C.c_void_p), ("splitk_ws_bytes", C.c_size_t), ("wt_frag", C.c_void_p)]


# every symbol include/yolo4hip.h declares: name -> (restype, argtypes)
_VP, _I, _F = C.c_void_p, C.c_int, C.c_float
SYMBOLS = {
    "y4_last_error": (C.c_char_p, []),
    "y4_version": (C.c_char_p, []),
    "y4_create": (_I, [C.POINTER(y4_config), C.POINTER(_VP)]),
    "y4_destroy": (_I, [_VP]),
    "y4_num_layers": (_I, [_VP]),
    "y4_layer_info": (_I, [_VP, _I, C.POINTER(y4_layer_desc)]),
    "y4_model_info": (_I, [_VP, C.POINTER(C.c_int64), C.POINTER(C.c_int32), C.POINTER(C.c_int32),
                           C.POINTER(C.c_int64)]),
    "y4_workspace_bytes": (_I, [_VP, C.POINTER(C.c_size_t), C.POINTER(C.c_size_t)]),
    "y4_bind_workspace": (_I, [_VP, _VP, C.c_size_t, _VP, C.c_size_t]),
    "y4_pack_weights": (_I, [_VP, _VP, C.c_size_t, _VP]),
    "y4_adopt_packed_weights": (_I, [_VP]),
    "y4_forward": (_I, [_VP, _VP, _I, _VP]),
    "y4_forward_u8": (_I, [_VP, _VP, _I, _VP]),
    "y4_forward_until": (_I, [_VP, _VP, _I, _I, _VP]),
    "y4_get_heads": (_I, [_VP, _I, _VP, _VP, _VP, _VP]),
    "y4_set_heads": (_I, [_VP, _I, _VP, _VP, _VP, _VP]),
    "y4_get_conv_output": (_I, [_VP, _I, _I, _VP, C.c_size_t, _VP]),
    "y4_decode_nms": (_I, [_VP, _I, _F, _F, _VP, _VP, _VP, _VP, _VP, _VP]),
    "y4_predict": (_I, [_VP, _VP, _I, _VP, _VP, _VP, _VP, _VP, _VP]),
    "y4_predict_u8": (_I, [_VP, _VP, _I, _VP, _VP, _VP, _VP, _VP, _VP]),
    "y4_profile": (_I, [_VP, _VP, _I, _VP, _VP, _I, C.POINTER(_I), _VP]),
    "y4_set_workspace_aliasing": (_I, [_VP, _I]),
    "y4_autotune": (_I, [_VP, _I, _I, _VP]),
    "y4_autotune_pair": (_I, [_VP, _VP, _I, _I, _VP, _VP, _I]),
    "y4_get_tiles": (_I, [_VP, C.POINTER(C.c_int32), _I]),
    "y4_set_tiles": (_I, [_VP, C.POINTER(C.c_int32), _I]),
    "y4_copy_schedule": (_I, [_VP, _VP]),
    "y4_set_subbatch": (_I, [_VP, _I, _I]),
    "y4_set_stem_fusion": (_I, [_VP, _I]),
    "y4_set_chain_fusion": (_I, [_VP, _I]),
    "y4_set_stage_fusion": (_I, [_VP, _I]),
    "y4_get_stage_fusion": (_I, [_VP]),
    "y4_set_res_fusion": (_I, [_VP, _I]),
    "y4_get_res_fusion": (_I, [_VP]),
    "y4_set_res_fusion_mask": (_I, [_VP, _I]),
    "y4_launch_counts": (_I, [_VP, C.POINTER(C.c_int32), C.POINTER(C.c_int32)]),
    "y4_timing_begin": (_I, [_VP, _I, _I]),
    "y4_timing_end": (_I, [_VP, _VP, _VP, _I, C.POINTER(_I), C.POINTER(_I), _VP]),
    "y4_packed_conv_bytes": (_I, [_I, _I, _I, _I, C.POINTER(C.c_int32), C.POINTER(C.c_size_t)]),
    "y4_pack_conv_weights": (_I, [_I, _I, _I, _I, _VP, _VP, _VP]),
    "y4_pack_conv_frag32": (_I, [_I, _I, _I, _VP, _VP, _VP]),
    "y4_conv2d": (_I, [C.POINTER(y4_conv_desc), _VP]),
    "y4_conv_tile_count": (_I, []),
    "y4_conv_tile_desc": (_I, [_I, C.POINTER(C.c_int32)]),
    "y4_set_splitk": (_I, [_VP, _I]),
    "y4_set_halo2": (_I, [_VP, _I]),
    "y4_pack_stem_weights": (_I, [_VP, _VP, _I, _VP]),
    "y4_stem_conv": (_I, [_I, _VP, _I, _I, _I, _VP, _VP, _VP, _I, _I, _VP, _I, _I, _VP]),
    "y4_preprocess_u8": (_I, [_VP, _I, _I, _VP, _I, _I, _VP]),
    "y4_resize_u8": (_I, [_VP, _I, _I, _I, _VP, _I, _I, _VP]),
    "y4_spp": (_I, [_I, _VP, _I, _I, _I, _VP]),
}

_lib = None


def load():
    """dlopen the in-tree library; raises (never falls back) when it is missing."""
    global _lib
    if _lib is None:
        if not os.path.exists(LIB_PATH):
            raise ImportError(f"{LIB_PATH} not found: build it with `python __graft_entry__.py` "
                              f"(or yolo-v4-tf.keras_amd/csrc/build.sh); there is no CPU fallback")
        # torch first: PyTorch-ROCm bundles its own libamdhip64; if this library were loaded before it, the process would
        # hold two HIP runtimes (the system one pulled in here, torch's later) and the one this library talks to then
        # reports "no ROCm-capable device" while torch.cuda works -- seen when build() and smoke() share a process
        import torch  # noqa: F401
        lib = C.CDLL(LIB_PATH)
        for name, (res, args) in SYMBOLS.items():
            fn = getattr(lib, name)          # AttributeError if the .so does not export it
            fn.restype, fn.argtypes = res, args
        _lib = lib
    return _lib


def check(rc):
    if rc != 0:
        raise Y4Error(rc, load().y4_last_error().decode("utf-8", "replace"))
    return rc


def stream_ptr(stream=None):
    import torch
    s = stream if stream is not None else torch.cuda.current_stream()
    return C.c_void_p(s.cuda_stream)


def ptr(t):
    return C.c_void_p(t.data_ptr()) if t is not None else C.c_void_p(0)
