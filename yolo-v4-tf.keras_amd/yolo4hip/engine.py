"""Engine: one libyolo4hip handle on one GPU, with its workspaces held as torch-ROCm tensors.

Stands where TensorFlow's runtime stands in the reference: `Yolov4.yolo_model.predict`
and `Yolov4.inference_model.predict` (reference models.py:113,159,514) end up in `Engine.forward_heads`
/ `Engine.predict`.  No compute happens in Python and nothing here falls back to a CPU path.
"""
import ctypes as C

import numpy as np

from . import ext
from .config import yolo_config


def _cfg_struct(config, num_classes, max_batch, dtype):
    c = ext.y4_config()
    size = config["img_size"]
    assert size[0] == size[1], "not support yet"                      # reference models.py:23
    assert size[0] % config["strides"][-1] == 0, "must be a multiple of last stride"   # models.py:24
    assert num_classes > 0, "no classes detected!"                   # models.py:38
    c.img_size = int(size[0]); c.num_classes = int(num_classes); c.max_batch = int(max_batch)
    c.dtype = ext.DTYPE_IDS[dtype] if isinstance(dtype, str) else int(dtype)
    anchors = np.asarray(config["anchors"], dtype=np.float32).reshape(-1)
    assert anchors.size == 18, "9 anchors (w,h) expected"
    for i in range(18):
        c.anchors[i] = float(anchors[i])
    for i in range(3):
        c.xyscale[i] = float(config["xyscale"][i])
        c.strides[i] = int(config["strides"][i])
    c.iou_threshold = float(config["iou_threshold"])
    c.score_threshold = float(config["score_threshold"])
    c.max_per_class = 100                       # custom_layers.py:293 (hard-coded in the reference)
    c.max_total = 100                           # custom_layers.py:294
    return c


class Engine:
    def __init__(self, num_classes, config=None, max_batch=32, dtype="f32", device=None, alias_workspace=False):
        import torch
        self.torch = torch
        self.lib = ext.load()
        if not torch.cuda.is_available():
            raise RuntimeError("yolo4hip needs a ROCm GPU (torch.cuda.is_available() is False); "
                               "there is no CPU fallback")
        self.config = dict(config or yolo_config)
        self.device = torch.device(device if device is not None else f"cuda:{torch.cuda.current_device()}")
        self.num_classes = int(num_classes)
        self.max_batch = int(max_batch)
        self.cfg = _cfg_struct(self.config, num_classes, max_batch, dtype)
        self.dtype = ext.DTYPE_NAMES[self.cfg.dtype]
        self.img_size = self.cfg.img_size
        self.handle = C.c_void_p()
        ext.check(self.lib.y4_create(C.byref(self.cfg), C.byref(self.handle)))
        flops, nbox, hcs, wfl = C.c_int64(), C.c_int32(), C.c_int32(), C.c_int64()
        ext.check(self.lib.y4_model_info(self.handle, C.byref(flops), C.byref(nbox), C.byref(hcs), C.byref(wfl)))
        self.flops_per_image, self.num_boxes = flops.value, nbox.value
        self.head_cstride, self.weight_floats = hcs.value, wfl.value
        # alias_workspace: activation buffers with disjoint lifetimes share memory (y4_set_workspace_aliasing): ~4x less
        # activation memory; conv_output() taps are then unavailable
        self.alias_workspace = bool(alias_workspace)
        if self.alias_workspace:
            ext.check(self.lib.y4_set_workspace_aliasing(self.handle, 1))
        a, w = C.c_size_t(), C.c_size_t()
        ext.check(self.lib.y4_workspace_bytes(self.handle, C.byref(a), C.byref(w)))
        self.act_bytes, self.wts_bytes = a.value, w.value
        with torch.cuda.device(self.device):
            self.act = torch.empty(self.act_bytes, dtype=torch.uint8, device=self.device)
            self.wts = torch.zeros(self.wts_bytes, dtype=torch.uint8, device=self.device)
            ext.check(self.lib.y4_bind_workspace(self.handle, ext.ptr(self.act), self.act_bytes, ext.ptr(self.wts),
                                                 self.wts_bytes))
        self.grids = [self.img_size // s for s in self.config["strides"]]
        self.nout = 3 * (self.num_classes + 5)
        self.T = self.cfg.max_total
        self.halo2 = False
        import os
        if self.dtype != "f32" and os.environ.get("YOLO4HIP_HALO2", "0") == "1":
            self.set_halo2(True)          # (experiments: `autotune` may pick the halo2 tiles; shipped schedules that hold them need no switch)

    def sibling(self):
        """A second handle on the same GPU that SHARES this engine's packed weights (read-only) and owns a second activation
        workspace, with the same scheduling choices (fusions, tuned tiles).  Two independent batches can then be in flight
        on two HIP streams (`InFlight`): one batch's partial last rounds, its 32-workgroup NMS and its small 19^2 layers
        overlap the other batch's kernels.  Results are those of this engine, bit for bit."""
        torch = self.torch
        e = Engine.__new__(Engine)
        e.torch, e.lib, e.config, e.device = torch, self.lib, dict(self.config), self.device
        e.num_classes, e.max_batch, e.cfg, e.dtype, e.img_size = self.num_classes, self.max_batch, self.cfg, self.dtype, self.img_size
        e.handle = C.c_void_p()
        ext.check(self.lib.y4_create(C.byref(e.cfg), C.byref(e.handle)))
        e.alias_workspace = getattr(self, "alias_workspace", False)
        if e.alias_workspace:
            ext.check(self.lib.y4_set_workspace_aliasing(e.handle, 1))
        e.flops_per_image, e.num_boxes = self.flops_per_image, self.num_boxes
        e.head_cstride, e.weight_floats = self.head_cstride, self.weight_floats
        e.act_bytes, e.wts_bytes = self.act_bytes, self.wts_bytes
        with torch.cuda.device(self.device):
            e.act = torch.empty(e.act_bytes, dtype=torch.uint8, device=self.device)
            e.wts = self.wts                                   # the SAME tensor: packed weights are never written after packing
            ext.check(self.lib.y4_bind_workspace(e.handle, ext.ptr(e.act), e.act_bytes, ext.ptr(e.wts), e.wts_bytes))
        e.grids, e.nout, e.T = list(self.grids), self.nout, self.T
        e.adopt_packed()
        self.copy_schedule_to(e)
        return e

    def copy_schedule_to(self, e):
        """Give engine `e` (a sibling) this engine's scheduling choices: sub-batching, fusions, every run's state, tuned tiles
        (y4_copy_schedule mirrors the handle itself; a get_tiles -> set_tiles hop could not carry a run head's two tiles)."""
        ext.check(self.lib.y4_copy_schedule(self.handle, e.handle))
        e._subbatch = getattr(self, "_subbatch", None)
        e.stem_fusion = bool(getattr(self, "stem_fusion", False))
        e.chain_fusion = bool(getattr(self, "chain_fusion", False))

    def close(self):
        for e in getattr(self, "_stream_siblings", []):
            e.close()
        self._stream_siblings = []
        self._stream_slots = None                          # predict_stream's pinned staging buffers
        if getattr(self, "handle", None) is not None and self.handle:
            self.lib.y4_destroy(self.handle)
            self.handle = C.c_void_p()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    # ---------------------------------------------------------------- weights
    def load_weight_blob(self, flat):
        """flat: float32 Darknet stream (numpy or torch; see weights.flatten / read_darknet)."""
        torch = self.torch
        if isinstance(flat, np.ndarray):
            flat = torch.from_numpy(np.ascontiguousarray(flat, dtype=np.float32))
        flat = flat.to(self.device, dtype=torch.float32).contiguous()
        with torch.cuda.device(self.device):
            ext.check(self.lib.y4_pack_weights(self.handle, ext.ptr(flat), flat.numel(), ext.stream_ptr()))
            torch.cuda.current_stream().synchronize()      # `flat` may be freed after this returns

    # packed-weight cache: the packed workspace is a pure function of (weights, classes, dtype,
    # library layout version), so it can be written once and re-loaded without re-reading / re-packing the
    # 258 MB Darknet file.
    def save_packed(self, path):
        import json
        meta = {"version": self.lib.y4_version().decode(), "num_classes": self.num_classes, "dtype": self.dtype,
                "wts_bytes": self.wts_bytes}
        blob = self.wts.cpu().numpy()
        with open(path, "wb") as f:
            head = json.dumps(meta).encode()
            f.write(len(head).to_bytes(8, "little")); f.write(head); blob.tofile(f)

    def load_packed(self, path):
        import json
        with open(path, "rb") as f:
            n = int.from_bytes(f.read(8), "little")
            meta = json.loads(f.read(n).decode())
            want = {"version": self.lib.y4_version().decode(), "num_classes": self.num_classes, "dtype": self.dtype,
                    "wts_bytes": self.wts_bytes}
            if meta != want:
                raise ValueError(f"packed-weight cache {path} was written for {meta}, this engine needs {want}")
            blob = np.fromfile(f, dtype=np.uint8)
        if blob.size != self.wts_bytes:
            raise ValueError(f"packed-weight cache {path} is truncated: {blob.size} of {self.wts_bytes} bytes")
        self.wts.copy_(self.torch.from_numpy(blob).to(self.device))
        self.torch.cuda.synchronize(self.device)
        self.adopt_packed()

    def adopt_packed(self):
        ext.check(self.lib.y4_adopt_packed_weights(self.handle))

    def layer_table(self):
        n = self.lib.y4_num_layers(self.handle)
        out = []
        for i in range(n):
            d = ext.y4_layer_desc()
            ext.check(self.lib.y4_layer_info(self.handle, i, C.byref(d)))
            out.append({k: getattr(d, k) for k, _ in ext.y4_layer_desc._fields_})
        return out

    # ---------------------------------------------------------------- compute
    def _to_device_images(self, imgs):
        """array-like [N,H,W,3] any float dtype (Keras casts to float32) -> contiguous cuda float32.  uint8 input -- a
        torch tensor OR a numpy array, the container must not change the meaning -- is "frames at network size before the
        /255" (what `preprocess_u8` returns, also after a `.cpu().numpy()` round trip): it stays uint8 and the stem
        divides inside its operand load.  Other integer dtypes are refused: their scale would be a guess."""
        torch = self.torch
        if isinstance(imgs, torch.Tensor) and imgs.dtype == torch.uint8:
            t = imgs.to(self.device)                        # network-size uint8 frames: the stem divides by 255
        elif isinstance(imgs, torch.Tensor):
            if not imgs.dtype.is_floating_point:
                raise ValueError(f"images must be floating point in [0,1] or uint8 frames, got {imgs.dtype}")
            t = imgs.to(self.device, dtype=torch.float32)
        else:
            a = np.asarray(imgs)
            if a.dtype == np.uint8:
                t = torch.from_numpy(np.ascontiguousarray(a)).to(self.device)
            elif a.dtype.kind != "f":
                raise ValueError(f"images must be floating point in [0,1] or uint8 frames, got {a.dtype}")
            else:
                if a.dtype != np.float32:
                    a = a.astype(np.float32)
                t = torch.from_numpy(np.ascontiguousarray(a)).to(self.device)
        if t.dim() != 4 or t.shape[1] != self.img_size or t.shape[2] != self.img_size or t.shape[3] != 3:
            raise ValueError(f"expected images of shape [N,{self.img_size},{self.img_size},3], got {tuple(t.shape)}")
        return t.contiguous()

    def preprocess_u8(self, raw_imgs, as_float=False):
        """Device-side `Yolov4.preprocess_img` (reference models.py:95-98) for one uint8 RGB image [h,w,3] or a list of
        them (any sizes).  Returns the tensor `forward_device` / `predict` take: by default a uint8 cuda tensor [n,S,S,3]
        (the resize done with cv2's uint8 arithmetic; the `/ 255.` then happens inside the stem's operand load), with
        as_float=True the float32 tensor in [0,1] (`y4_preprocess_u8`).  Both give bit-identical network outputs."""
        torch = self.torch
        if isinstance(raw_imgs, np.ndarray) and raw_imgs.ndim == 3:
            raw_imgs = [raw_imgs]
        S = self.img_size
        out = torch.empty((len(raw_imgs), S, S, 3), dtype=torch.float32 if as_float else torch.uint8, device=self.device)
        with torch.cuda.device(self.device):
            for i, im in enumerate(raw_imgs):
                a = np.ascontiguousarray(im)
                if a.dtype != np.uint8 or a.ndim != 3 or a.shape[2] != 3:
                    raise ValueError(f"expected uint8 [h,w,3] images, got {a.dtype} {a.shape}")
                d = torch.from_numpy(a.copy() if not a.flags.writeable else a).to(self.device)
                if as_float:
                    ext.check(self.lib.y4_preprocess_u8(ext.ptr(d), a.shape[0], a.shape[1], ext.ptr(out[i]), S, S, ext.stream_ptr()))
                else:
                    ext.check(self.lib.y4_resize_u8(ext.ptr(d), 1, a.shape[0], a.shape[1], ext.ptr(out[i]), S, S, ext.stream_ptr()))
            torch.cuda.current_stream().synchronize()       # the uint8 staging tensors may be freed now
        return out

    def resize_u8(self, frames_dev, out=None):
        """Device-side `cv2.resize(img, img_size)` on a uint8 cuda batch [n,h,w,3] -> uint8 [n,S,S,3] (cv2's uint8
        INTER_LINEAR arithmetic); frames already at network size are returned as they are."""
        torch = self.torch
        n, h, w, _ = frames_dev.shape
        S = self.img_size
        if h == S and w == S:
            return frames_dev
        if out is None:
            out = torch.empty((n, S, S, 3), dtype=torch.uint8, device=self.device)
        with torch.cuda.device(self.device):
            ext.check(self.lib.y4_resize_u8(ext.ptr(frames_dev), n, h, w, ext.ptr(out), S, S, ext.stream_ptr()))
        return out[:n]

    def _check_device_images(self, imgs_dev):
        torch = self.torch
        S = self.img_size
        if imgs_dev.dtype not in (torch.float32, torch.uint8) or imgs_dev.dim() != 4 or tuple(imgs_dev.shape[1:]) != (S, S, 3) \
                or not imgs_dev.is_contiguous():
            raise ValueError(f"expected a contiguous float32 or uint8 cuda tensor [n,{S},{S},3], got {imgs_dev.dtype} {tuple(imgs_dev.shape)}")
        return imgs_dev.dtype == torch.uint8

    def forward_device(self, imgs_dev):
        """imgs_dev: float32 [n,S,S,3] in [0,1] (what preprocess_img gives), or uint8 [n,S,S,3] frames at network size
        BEFORE the /255 -- the stem then divides inside its operand load (bit-identical, no float image tensor)."""
        n = imgs_dev.shape[0]
        u8 = self._check_device_images(imgs_dev)
        with self.torch.cuda.device(self.device):
            fn = self.lib.y4_forward_u8 if u8 else self.lib.y4_forward
            ext.check(fn(self.handle, ext.ptr(imgs_dev), n, ext.stream_ptr()))

    def forward_until_device(self, imgs_dev, last_conv=71):
        """The forward pass cut behind conv `last_conv` (default 71: CSPDarknet53 proper, reference custom_layers.py:100-124) on a
        float32 device batch -- what `bench.py` times as `backbone`.  Asynchronous on the current stream."""
        n = imgs_dev.shape[0]
        if self._check_device_images(imgs_dev):
            raise ValueError("forward_until_device takes float32 images")
        with self.torch.cuda.device(self.device):
            ext.check(self.lib.y4_forward_until(self.handle, ext.ptr(imgs_dev), n, int(last_conv), ext.stream_ptr()))

    def heads_device(self, n):
        torch = self.torch
        outs = [torch.empty((n, g, g, self.nout), dtype=torch.float32, device=self.device) for g in self.grids]
        with torch.cuda.device(self.device):
            ext.check(self.lib.y4_get_heads(self.handle, n, ext.ptr(outs[0]), ext.ptr(outs[1]), ext.ptr(outs[2]),
                                            ext.stream_ptr()))
        return outs

    def set_heads(self, heads):
        """Load dense float32 raw heads (3 arrays [n,g,g,3(C+5)]) into the workspace (decode/NMS tests)."""
        torch = self.torch
        ts = [torch.from_numpy(np.ascontiguousarray(h, dtype=np.float32)).to(self.device) for h in heads]
        n = ts[0].shape[0]
        for t, g in zip(ts, self.grids):
            if tuple(t.shape) != (n, g, g, self.nout):
                raise ValueError(f"head shape {tuple(t.shape)} != {(n, g, g, self.nout)}")
        with torch.cuda.device(self.device):
            ext.check(self.lib.y4_set_heads(self.handle, n, ext.ptr(ts[0]), ext.ptr(ts[1]), ext.ptr(ts[2]),
                                            ext.stream_ptr()))
            torch.cuda.current_stream().synchronize()
        return n

    def alloc_outputs(self, n):
        torch = self.torch
        T = self.T
        return (torch.empty((n, T, 4), dtype=torch.float32, device=self.device),
                torch.empty((n, T), dtype=torch.float32, device=self.device),
                torch.empty((n, T), dtype=torch.float32, device=self.device),
                torch.empty((n,), dtype=torch.int32, device=self.device),
                torch.empty((n, T), dtype=torch.int32, device=self.device))

    def alloc_outputs_flat(self, n):
        """The five outputs of `alloc_outputs` as views of ONE int32 device block, so that a step's results return to
        the host in a single copy: -> (flat int32 tensor, (boxes, scores, classes, valid, kept) views)."""
        torch = self.torch
        T = self.T
        sizes = [n * T * 4, n * T, n * T, n, n * T]
        offs = np.cumsum([0] + sizes)
        flat = torch.empty(int(offs[-1]), dtype=torch.int32, device=self.device)
        v = [flat[offs[i]:offs[i + 1]] for i in range(5)]
        return flat, (v[0].view(torch.float32).view(n, T, 4), v[1].view(torch.float32).view(n, T),
                      v[2].view(torch.float32).view(n, T), v[3], v[4].view(n, T))

    def decode_nms_device(self, n, outs=None, iou_threshold=-1.0, score_threshold=-1.0):
        outs = outs or self.alloc_outputs(n)
        b, s, c, v, k = outs
        with self.torch.cuda.device(self.device):
            ext.check(self.lib.y4_decode_nms(self.handle, n, float(iou_threshold), float(score_threshold), ext.ptr(b),
                                             ext.ptr(s), ext.ptr(c), ext.ptr(v), ext.ptr(k), ext.stream_ptr()))
        return outs

    def predict_device(self, imgs_dev, outs=None):
        """The whole hot path on device buffers: forward + decode + NMS (async on the current stream)."""
        n = imgs_dev.shape[0]
        u8 = self._check_device_images(imgs_dev)
        outs = outs or self.alloc_outputs(n)
        b, s, c, v, k = outs
        with self.torch.cuda.device(self.device):
            fn = self.lib.y4_predict_u8 if u8 else self.lib.y4_predict
            ext.check(fn(self.handle, ext.ptr(imgs_dev), n, ext.ptr(b), ext.ptr(s), ext.ptr(c),
                         ext.ptr(v), ext.ptr(k), ext.stream_ptr()))
        return outs

    def _chunks(self, imgs):
        t = self._to_device_images(imgs)
        if t.shape[0] == 0:                                 # Keras' Model.predict raises ValueError on an empty input as well
            raise ValueError("predict: no images (input of shape %s)" % (tuple(t.shape),))
        for i in range(0, t.shape[0], self.max_batch):     # Keras predict() chunks too (batch_size=32), invisibly
            yield t[i:i + self.max_batch]

    def forward_heads(self, imgs):
        """yolo_model.predict(imgs): list of 3 float32 numpy arrays [N,g,g,3(C+5)] (raw logits)."""
        parts = [[], [], []]
        for chunk in self._chunks(imgs):
            self.forward_device(chunk)
            for i, o in enumerate(self.heads_device(chunk.shape[0])):
                parts[i].append(o.cpu().numpy())
        return [np.concatenate(p, axis=0) for p in parts]

    def predict(self, imgs, with_indices=False, iou_threshold=-1.0, score_threshold=-1.0):
        """inference_model.predict(imgs): [boxes [N,100,4], scores [N,100], classes [N,100], valid [N] int32]
        as fresh, writable host numpy arrays (the reference's export_prediction mutates them in place,
        models.py:167-168)."""
        acc = [[], [], [], [], []]
        for chunk in self._chunks(imgs):
            self.forward_device(chunk)
            outs = self.decode_nms_device(chunk.shape[0], None, iou_threshold, score_threshold)
            for i, o in enumerate(outs):
                acc[i].append(o.cpu().numpy())
        res = [np.concatenate(p, axis=0) for p in acc]
        return res if with_indices else res[:4]

    def predict_stream(self, batches, with_indices=False, in_flight=None):
        """Pipelined `inference_model.predict` over an iterable of uint8 batches ([n,h,w,3] numpy arrays or pinned torch
        tensors, n <= max_batch, any h,w): yields one result list per batch, in order.  A pinned tensor is uploaded from where
        it lies and may be refilled as soon as the generator yields (its upload is waited for before every yield).  While batch i computes, batch i+1 crosses PCIe as uint8
        on a second HIP stream (pinned staging, 4x fewer bytes than float32) and batch i-1's results return to the
        host, so the PCIe-inclusive rate approaches the device rate.  No float image tensor exists: frames are resized
        uint8 -> uint8 on the device when needed (`y4_resize_u8`) and the `/ 255.` happens inside the stem's operand load
        (`y4_predict_u8`), bit-identical to `Yolov4.preprocess_img` (reference models.py:95-98) + float32 forward.
        `in_flight` batches compute at the same time, each on its own HIP stream and activation workspace (`sibling()`: shared
        packed weights), so that one batch's idle compute units are the other's (see `InFlight`); results are unchanged and still
        come in order.  Every batch in flight beyond the first costs one more activation workspace for the engine's lifetime
        (`act_bytes`: 2.9 GB at 608x608 / batch 32 / bf16 with `alias_workspace`, 8.0 GB without), so the default is 2 with an
        aliased workspace and 1 with the plain one; pass `in_flight` to choose."""
        torch = self.torch
        dev = self.device
        copy_stream = torch.cuda.Stream(device=dev)        # uploads
        down_stream = torch.cuda.Stream(device=dev)        # results (its own stream: it waits for the compute stream)
        if in_flight is None:
            in_flight = 2 if self.alias_workspace else 1
        in_flight = max(1, int(in_flight))
        sibs = getattr(self, "_stream_siblings", [])
        while len(sibs) < in_flight - 1:
            sibs.append(self.sibling())
        self._stream_siblings = sibs
        for e in sibs[:in_flight - 1]:
            self.copy_schedule_to(e)                       # (the engine may have been re-tuned since the sibling was made)
        engines = [self] + sibs[:in_flight - 1]
        nslots = in_flight + 1                             # one more staging slot than computing batches: the next upload runs ahead
        # the staging slots (pinned host buffers, device frames, output blocks, events) outlive the call: pinning tens of MB
        # costs milliseconds, which a short stream would pay again on every call
        cache = getattr(self, "_stream_slots", None)
        if cache is None or len(cache) != nslots:
            for sl in cache or []:                         # an abandoned generator may have left copies in flight on these buffers
                for ev in ("up", "ran", "done"):
                    if ev in sl:
                        sl[ev].synchronize()
            cache = self._stream_slots = [{} for _ in range(nslots)]
        slots = cache
        for sl in slots:
            if "done" in sl:
                sl["done"].synchronize()
        pending = []

        T, nb = self.T, self.max_batch
        # one flat int32 block per slot holds all five outputs, so that results return in ONE small D2H copy
        sizes = [nb * T * 4, nb * T, nb * T, nb, nb * T]
        offs = np.cumsum([0] + sizes)

        def views(flat):
            v = [flat[offs[i]:offs[i + 1]] for i in range(5)]
            return (v[0].view(torch.float32).view(nb, T, 4), v[1].view(torch.float32).view(nb, T),
                    v[2].view(torch.float32).view(nb, T), v[3], v[4].view(nb, T))

        def finish(sl):
            sl["done"].synchronize()
            res = [t[:sl["n"]].numpy().copy() for t in sl["host"]]
            return res if with_indices else res[:4]

        with torch.cuda.device(dev):
            cstreams = [torch.cuda.current_stream()] + [torch.cuda.Stream(device=dev) for _ in engines[1:]]
            for bi, batch in enumerate(batches):
                pinned_in = isinstance(batch, torch.Tensor) and batch.is_pinned() and batch.is_contiguous()
                a = batch if pinned_in else np.ascontiguousarray(batch)
                if a.dtype != (torch.uint8 if pinned_in else np.uint8) or a.ndim != 4 or a.shape[3] != 3 or not 1 <= a.shape[0] <= self.max_batch:
                    raise ValueError(f"expected uint8 [n<={self.max_batch},h,w,3] batches, got {a.dtype} {a.shape}")
                sl = slots[bi % nslots]
                eng, compute = engines[bi % in_flight], cstreams[bi % in_flight]     # batches alternate engines; slots rotate on their own
                shape = tuple(a.shape)
                if sl.get("shape") != shape:               # (re)allocate this slot's staging for the frame geometry
                    if "done" in sl:
                        sl["done"].synchronize()
                    sl["shape"] = shape
                    sl["pin"] = torch.empty(shape, dtype=torch.uint8).pin_memory()
                    sl["pin_np"] = sl["pin"].numpy()
                    sl["u8"] = torch.empty(shape, dtype=torch.uint8, device=dev)
                    # frames of another size are resized uint8 -> uint8 on the device; the /255 happens in the stem's load
                    sl["net"] = None if shape[1:3] == (self.img_size, self.img_size) else \
                        torch.empty((self.max_batch, self.img_size, self.img_size, 3), dtype=torch.uint8, device=dev)
                    sl["flat"] = torch.empty(int(offs[-1]), dtype=torch.int32, device=dev)
                    sl["flat_host"] = torch.empty(int(offs[-1]), dtype=torch.int32).pin_memory()
                    sl["outs"], sl["host"] = views(sl["flat"]), views(sl["flat_host"])
                    sl["up"], sl["done"], sl["free"], sl["ran"] = (torch.cuda.Event() for _ in range(4))
                    sl["free"].record(compute)
                n = a.shape[0]
                sl["n"] = n
                sl["free"].synchronize()                   # the slot's previous uint8 frame has been consumed
                # plain single-threaded memcpy into the pinned staging buffer: torch's multi-threaded host copy leaves
                # its worker pool spinning, which stalls the HIP runtime's own threads for tens of ms (measured).
                # A batch that already is a pinned uint8 tensor is uploaded from where it lies.
                if not pinned_in:
                    np.copyto(sl["pin_np"], a)
                with torch.cuda.stream(copy_stream):
                    sl["u8"].copy_(batch if pinned_in else sl["pin"], non_blocking=True)
                    sl["up"].record(copy_stream)
                compute.wait_event(sl["up"])
                with torch.cuda.stream(compute):
                    frames = sl["u8"] if sl["net"] is None else eng.resize_u8(sl["u8"], sl["net"])
                    eng.predict_device(frames[:n], tuple(t[:n] for t in sl["outs"]))
                sl["free"].record(compute)                 # the stem has consumed the uint8 frames
                sl["ran"].record(compute)
                with torch.cuda.stream(down_stream):       # results leave on a third stream: compute and uploads never wait
                    down_stream.wait_event(sl["ran"])
                    sl["flat_host"].copy_(sl["flat"], non_blocking=True)
                    sl["done"].record(down_stream)
                if pinned_in:
                    # the upload reads the CALLER's pinned tensor: it must have left host memory before control returns
                    # to a loader that may refill that buffer (the engine's own staging buffer is guarded by sl["free"])
                    sl["up"].synchronize()
                pending.append(sl)
                if len(pending) >= nslots:                 # the oldest batch's slot is needed next: its results leave first
                    yield finish(pending.pop(0))
            while pending:
                yield finish(pending.pop(0))

    def conv_output(self, conv_idx, n):
        """Dense float32 NHWC copy of conv `conv_idx`'s output from the last forward (parity tap)."""
        torch = self.torch
        lt = self.layer_table()[conv_idx]
        # upsampling convs (78, 85) store the 2x-upsampled tensor
        side = lt["out_side"] * (2 if conv_idx in (78, 85) else 1)
        out = torch.empty((n, side, side, lt["cout"]), dtype=torch.float32, device=self.device)
        with torch.cuda.device(self.device):
            ext.check(self.lib.y4_get_conv_output(self.handle, conv_idx, n, ext.ptr(out), out.numel(), ext.stream_ptr()))
        return out.cpu().numpy()

    def get_tiles(self):
        """The 110 tile ids this engine runs (y4_get_tiles)."""
        tiles = (C.c_int32 * 110)()
        ext.check(self.lib.y4_get_tiles(self.handle, tiles, 110))
        return list(tiles)

    def autotune(self, n=None, reps=3):
        """Pick the fastest tile configuration per conv layer by measurement (results are bit-identical)."""
        n = int(n or self.max_batch)
        with self.torch.cuda.device(self.device):
            ext.check(self.lib.y4_autotune(self.handle, n, int(reps), ext.stream_ptr()))
        return self.get_tiles()

    def autotune_pair(self, other, stream, other_stream, n=None, reps=3, passes=15):
        """`autotune` with throughput as objective, for two batches in flight (`InFlight`): every candidate is timed on this
        engine and on its sibling `other` at once, each on its own HIP stream (y4_autotune_pair).  Both engines end up
        with the same choices; results stay bit-identical.  `passes`: which decisions use the two-stream objective -- bit 0
        tiles, 1 chains / LDS pairs, 2 the stage kernel, 3 the residual-block kernels (the rest: one launch at a time)."""
        n = int(n or self.max_batch)
        with self.torch.cuda.device(self.device):
            ext.check(self.lib.y4_autotune_pair(self.handle, other.handle, n, int(reps), C.c_void_p(stream.cuda_stream),
                                                C.c_void_p(other_stream.cuda_stream), int(passes)))
        return self.get_tiles()

    # ---------------------------------------------------------------- shipped schedules
    def shipped_schedule(self):
        """The tuned schedule that ships with the package for this (image side, classes, batch, dtype), or None: the
        per-launch tile ids, the stage-kernel switch and the residual-block mask one `autotune` run chose on an MI355X
        (`yolo4hip/schedules/<side>_<classes>_<batch>_<dtype>.json`; the headline shape's file IS `profiles/r03/tiles.json`,
        the set the committed PMC passes profiled).  A schedule WITHOUT split-K or halo2 ids (`"splitk": false`, `"halo2": false`) is a
        pure scheduling choice: every such choice gives the same bits.  The batch-1 files carry split-K ids (`"splitk": true`), the
        batch-32 / 64 16-bit files since round 6 halo2 ids (`"halo2": true`: tile ids 55-62, conv_halo2_kernel.h, v_mfma_32x32x16):
        those sum the K loop in another, fixed fp32 order -- the same on every machine because the file is the same -- and are held
        to the oracle instead (tests/test_gpu_forward.py::test_splitk_latency_schedule_vs_oracle, tests/test_gpu_parity_full.py)."""
        import json
        import os
        path = os.path.join(os.path.dirname(os.path.abspath(__file__)), "schedules",
                            f"{self.img_size}_{self.num_classes}_{self.max_batch}_{self.dtype}.json")
        try:
            saved = json.load(open(path))
        except (OSError, ValueError):
            return None
        saved["path"] = path
        return saved

    # ---------------------------------------------------------------- no silent un-tuned shapes
    LATENCY_BATCH = 2          # engines for at most this many images are tuned for latency: split-K tile ids allowed

    def schedule_cache_path(self):
        """Where a schedule tuned on first use is kept: $YOLO4HIP_CACHE (default ~/.cache/yolo4hip) / schedules /
        <side>_<classes>_<batch>_<dtype>_<gfx arch>_<library version>.json."""
        import os
        root = os.environ.get("YOLO4HIP_CACHE") or os.path.join(os.path.expanduser("~"), ".cache", "yolo4hip")
        arch = self.torch.cuda.get_device_properties(self.device).gcnArchName.split(":")[0]
        ver = self.lib.y4_version().decode().replace(" ", "_").replace("/", "_")
        return os.path.join(root, "schedules", f"{self.img_size}_{self.num_classes}_{self.max_batch}_{self.dtype}_{arch}_{ver}.json")

    def ensure_schedule(self, tune=True, verbose=True, share=False):
        """Make sure this engine runs a TUNED schedule, and say which: (1) the one that ships with the package for this (image side,
        classes, batch, dtype), else (2) the one a previous process tuned on this machine (`schedule_cache_path`), else (3) tune
        now -- all fusions on, `autotune` on a synthetic batch of `max_batch` images (a full-size predict + a few launches per
        candidate tile: seconds at 160^2, about 20 s at 608^2 x 32), split-K offered only to latency-sized engines
        (`max_batch <= LATENCY_BATCH`) that ask for it (YOLO4HIP_LATENCY=1) -- and write it to the cache.  The weights must be
        loaded.  Every schedule without split-K ids gives the same bits; one with them is tested against the oracle instead
        (`"splitk": true` in the file).  `share=True` under an initialised torch.distributed group makes the call a COLLECTIVE:
        rank 0 resolves (1)-(3), the others take rank 0's schedule from a broadcast (source 'shared') -- one tuning run per job,
        the same tile set on every rank.  EVERY rank of the group must make the call, with the same (size, classes,
        batch, dtype): the shape travels with the schedule and a mismatch raises; a rank 0 that fails to resolve broadcasts the
        failure, so the others raise instead of waiting for the process group's timeout.  The tuned schedule is judged one batch at a time (`"in_flight": 1`) even
        when `predict_stream` later keeps two in flight; `bench.py --pair-passes` / `y4_autotune_pair` is the pair-judged variant.
        Returns (source, path): source in 'shipped' | 'cached' | 'tuned' | 'shared' | 'heuristic' (tune=False and nothing found)."""
        from . import dist as D
        rank, world = D.group_rank_world() if share else (0, 1)
        src, path, saved, err = "heuristic", None, None, None
        mine = [int(self.img_size), int(self.num_classes), int(self.max_batch), str(self.dtype)]
        if rank == 0:
            try:
                src, path, saved = self._resolve_schedule(tune)
            except Exception as e:                     # (a rank 0 that raised before the broadcast would leave the others hanging in it)
                if world == 1:
                    raise
                err = f"{type(e).__name__}: {e}"
        if world > 1:
            got = D.share_schedule({"schedule": saved, "error": err, "shape": mine} if rank == 0 else None, src=0)
            if got is None or got.get("error"):
                raise RuntimeError("ensure_schedule(share=True): rank 0 could not resolve a schedule" +
                                   (f" ({got['error']})" if got else ""))
            if rank != 0:
                if list(got["shape"]) != mine:
                    raise RuntimeError(f"ensure_schedule(share=True): rank 0 resolved a schedule for (size, classes, batch, dtype) = "
                                       f"{tuple(got['shape'])}, this rank's engine is {tuple(mine)} -- every rank must build the same engine")
                saved = got["schedule"]
                if saved is not None:
                    self._use_schedule(saved)
                    src, path = "shared", None
        self.schedule_source = (src, path)
        if verbose:
            self.say_schedule()
        return src, path

    def say_schedule(self):
        import os
        src, path = self.schedule_source
        what = {"shipped": "shipped with the package", "cached": "tuned earlier on this machine", "tuned": "tuned now (first use of this shape)",
                "shared": "rank 0's, received by broadcast", "heuristic": "NONE: built-in tile heuristic, fusion kernels off"}[src]
        print(f"schedule: {what}" + (f" [{os.path.basename(path)}]" if path else ""))

    def _all_fusions_on(self):
        if self.dtype != "f32":
            if self.img_size <= 640:
                self.set_stem_fusion(True)
            self.set_chain_fusion(True)
            self.set_stage_fusion(True)
            self.set_res_fusion(True)

    def _use_schedule(self, saved):
        self._all_fusions_on()
        self.apply_schedule(saved)

    def _resolve_schedule(self, tune):
        """(source, path, schedule dict or None) of `ensure_schedule`'s steps (1)-(3) on THIS rank; the schedule is applied."""
        import json
        import os
        saved = self.shipped_schedule()
        if saved is not None and len(saved.get("tiles", [])) == 110:
            self._use_schedule(saved)
            return "shipped", saved["path"], saved
        path = self.schedule_cache_path()
        try:
            saved = json.load(open(path))
            if len(saved.get("tiles", [])) != 110:
                raise ValueError("stale")
            self._use_schedule(saved)
            return "cached", path, saved
        except (OSError, ValueError, ext.Y4Error):
            pass
        if not tune:
            return "heuristic", None, None
        torch = self.torch
        self._all_fusions_on()
        # split-K ids only on request: which of them win is decided by this machine's timing, and they change
        # the fp32 summation order -- a schedule tuned on first use would make low-order bits machine-dependent.  The
        # shipped batch-1 schedules have them (one fixed file, tested against the oracle); YOLO4HIP_LATENCY=1 opts in here.
        splitk = self.max_batch <= self.LATENCY_BATCH and os.environ.get("YOLO4HIP_LATENCY", "0") == "1"
        self.set_splitk(splitk)
        from . import weights as W
        imgs = torch.from_numpy(W.synth_images(self.max_batch, self.img_size, seed=0)).to(self.device)
        self.predict_device(imgs)                      # real activations in the workspace
        tiles = self.autotune(self.max_batch, reps=3)
        self.set_splitk(False)
        saved = {"size": self.img_size, "classes": self.num_classes, "batch": self.max_batch, "dtype": self.dtype,
                 "tiles": tiles, "stage_fusion": bool(self.stage_fusion_active()) if self.dtype != "f32" else False,
                 "res_fusion_mask": int(self.res_fusion_mask()) if self.dtype != "f32" else 0, "in_flight": 1,
                 "splitk": any(abs(t) % 1000 >= 100 or abs(t) // 1000 >= 100 for t in tiles),
                 "halo2": bool(getattr(self, "halo2", False)) and any(55 <= abs(t) % 100 <= 62 for t in tiles),
                 "tuned_on": torch.cuda.get_device_properties(self.device).gcnArchName}
        try:
            os.makedirs(os.path.dirname(path), exist_ok=True)
            tmp = path + f".{os.getpid()}.tmp"
            json.dump(saved, open(tmp, "w"))
            os.replace(tmp, path)
        except OSError:
            path = None                                # read-only home: tuned for this process only
        return "tuned", path, saved

    def apply_schedule(self, saved):
        """Tiles / stage kernel / residual-block mask from a schedule dict (`shipped_schedule`, `bench.py --save-tiles`).  The
        fusion switches themselves (stem, chains, stage, residual blocks) must already be on: a schedule only narrows them."""
        self.set_tiles(saved["tiles"])
        if self.dtype != "f32" and "stage_fusion" in saved and self.stage_fusion_active():
            self.set_stage_fusion(bool(saved["stage_fusion"]))
        if self.dtype != "f32" and "res_fusion_mask" in saved and self.res_fusion_mask():
            self.set_res_fusion_mask(int(saved["res_fusion_mask"]))

    def set_tiles(self, tiles):
        arr = (C.c_int32 * len(tiles))(*[int(t) for t in tiles])
        ext.check(self.lib.y4_set_tiles(self.handle, arr, len(tiles)))

    def set_splitk(self, on=True):
        """Latency schedules: let `autotune` also offer split-K tile ids (the K loop of a tile split over 2 / 4 / 8 workgroups; for
        small batches, where the deep layers have fewer tiles than the GPU has compute units).  A split launch adds its partial
        sums in another fp32 order than the unsplit loop, so the tuned schedule then is part of the numerical result; off by
        default, and the schedules that ship for batch 1 say so (`"splitk": true`)."""
        ext.check(self.lib.y4_set_splitk(self.handle, int(bool(on))))
        self.splitk = bool(on)

    def set_halo2(self, on=True):
        """Let `autotune` also offer the halo2 tiles (conv_halo2_kernel.h: one wave per SIMD, v_mfma_32x32x16, weights in registers)
        for the 3x3 stride-1 convs.  They sum the K axis in another fixed fp32 order than the 16x16x32 tiles, so -- as with split-K --
        a schedule that holds them (`"halo2": true`) is part of the numerical result and is tested against the oracle."""
        ext.check(self.lib.y4_set_halo2(self.handle, int(bool(on))))
        self.halo2 = bool(on)

    def set_subbatch(self, images, last_conv=16):
        """Run convs 0..last_conv over `images` images at a time (Infinity-Cache residency of the big early
        activations); 0 turns it off.  Results are unchanged."""
        ext.check(self.lib.y4_set_subbatch(self.handle, int(images), int(last_conv)))
        self._subbatch = (int(images), int(last_conv)) if images > 0 else None

    def set_stem_fusion(self, on=True):
        """Convs 0+1 as one kernel with conv 0's output kept in LDS (16-bit dtypes, img_size <= 640).  Results are
        unchanged; conv_output(0) is unavailable while on."""
        ext.check(self.lib.y4_set_stem_fusion(self.handle, int(bool(on))))
        self.stem_fusion = bool(on)

    def set_chain_fusion(self, on=True):
        """3x3+Add -> 1x1 (-> 1x1 over the concat) runs of the 64-channel CSP stages as one kernel each (16-bit
        dtypes).  Returns the number of fused runs.  A chained conv issues the same MFMAs on the same 16-bit inputs in the
        same order as its own kernel, so every materialised tensor, the heads and the detections are BIT-IDENTICAL to the
        unfused path (tests/test_gpu_forward.py::test_chain_fusion_is_bit_identical): a pure scheduling knob."""
        r = self.lib.y4_set_chain_fusion(self.handle, int(bool(on)))
        if r < 0:
            ext.check(r)
        self.chain_fusion = bool(on)
        return r

    def set_stage_fusion(self, on=True):
        """Convs 2..7 (the 304^2 CSP stage at 608) as one spatially tiled kernel (16-bit dtypes; bit-identical results).
        Returns whether the stage kernel is active; `autotune` afterwards keeps it only if it measures faster."""
        r = self.lib.y4_set_stage_fusion(self.handle, int(bool(on)))
        if r < 0:
            ext.check(r)
        return bool(r)

    def stage_fusion_active(self):
        r = self.lib.y4_get_stage_fusion(self.handle)
        if r < 0:
            ext.check(r)
        return bool(r)

    def set_res_fusion(self, on=True):
        """Residual blocks of the 64- / 128-channel stages (1x1 -> 3x3 + Add) as one spatially tiled kernel each (16-bit
        dtypes; bit-identical).  Returns the number of such blocks; `autotune` afterwards keeps them per channel group
        only where they measure faster (`res_fusion_mask`: bit 0 = 128 channels, bit 1 = 64 channels)."""
        r = self.lib.y4_set_res_fusion(self.handle, int(bool(on)))
        if r < 0:
            ext.check(r)
        return r

    def res_fusion_mask(self):
        r = self.lib.y4_get_res_fusion(self.handle)
        if r < 0:
            ext.check(r)
        return r

    def set_res_fusion_mask(self, mask):
        ext.check(self.lib.y4_set_res_fusion_mask(self.handle, int(mask)))

    def conv_launches_per_step(self):
        """Launches of the conv kernel family (everything but the stem) in one predict under the current settings."""
        convs, total = C.c_int32(), C.c_int32()
        ext.check(self.lib.y4_launch_counts(self.handle, C.byref(convs), C.byref(total)))
        return convs.value

    def timing_begin(self, max_steps, coarse=False):
        ext.check(self.lib.y4_timing_begin(self.handle, int(max_steps), int(bool(coarse))))

    def timing_end(self):
        """-> ([(op name, mean ms)], steps recorded); synchronises the current stream."""
        cap = 256
        ms = (C.c_float * cap)()
        names = C.create_string_buffer(16 * cap)
        nops, steps = C.c_int(), C.c_int()
        with self.torch.cuda.device(self.device):
            ext.check(self.lib.y4_timing_end(self.handle, ms, names, cap, C.byref(nops), C.byref(steps),
                                             ext.stream_ptr()))
        raw = names.raw
        ops = [(raw[16 * i:16 * i + 16].split(b"\0")[0].decode(), float(ms[i])) for i in range(nops.value)]
        return ops, steps.value

    def profile(self, imgs_dev):
        n = imgs_dev.shape[0]
        cap = 256
        ms = (C.c_float * cap)()
        names = C.create_string_buffer(16 * cap)
        nops = C.c_int()
        with self.torch.cuda.device(self.device):
            ext.check(self.lib.y4_profile(self.handle, ext.ptr(imgs_dev), n, ms, names, cap, C.byref(nops),
                                          ext.stream_ptr()))
        raw = names.raw
        return [(raw[16 * i:16 * i + 16].split(b"\0")[0].decode(), float(ms[i])) for i in range(nops.value)]


class InFlight:
    """`depth` batches in flight on one GPU: the engine and depth-1 siblings (shared packed weights, own activation
    workspaces), each on its own HIP stream, used round-robin.  A batch's kernels run in order on its stream; kernels of
    different batches overlap wherever one leaves compute units idle (partial last rounds of its tiles, the 32-workgroup
    NMS, the sub-one-round 19^2 layers).  Measured at 608x608 / 80 classes / batch 32 / bf16: 5.57 -> 5.03 ms per batch
    with two in flight (three: 5.11).  Outputs are bit-identical to the single-stream path (tests/test_gpu_api.py)."""

    def __init__(self, engine, depth=2):
        torch = engine.torch
        if depth < 1:
            raise ValueError("depth must be >= 1")
        self.engines = [engine] + [engine.sibling() for _ in range(depth - 1)]
        with torch.cuda.device(engine.device):
            self.streams = [torch.cuda.Stream(device=engine.device) for _ in range(depth)]
        self.torch, self.depth, self._next = torch, depth, 0

    def autotune(self, n=None, reps=3, passes=15):
        """Tune engine 0 and its first sibling TOGETHER, with two batches in flight as the objective of the decisions in
        `passes` (y4_autotune_pair), and copy the result to the other siblings.  Both engines need activations in their
        workspaces (one predict each)."""
        torch = self.torch
        if self.depth < 2:
            return self.engines[0].autotune(n, reps)
        torch.cuda.synchronize(self.engines[0].device)
        tiles = self.engines[0].autotune_pair(self.engines[1], self.streams[0], self.streams[1], n, reps, passes)
        torch.cuda.synchronize(self.engines[0].device)
        for e in self.engines[2:]:
            self.engines[0].copy_schedule_to(e)
        return tiles

    def submit(self, imgs_dev, outs, host=None, flat=None):
        """predict_device(imgs_dev, outs) on the next slot's stream (asynchronous); with `host` / `flat` the results are
        also copied to the pinned host tensor.  The caller owns per-slot inputs / outputs: slot = call index % depth."""
        k = self._next
        self._next = (k + 1) % self.depth
        with self.torch.cuda.stream(self.streams[k]):
            self.engines[k].predict_device(imgs_dev, outs)
            if host is not None:
                host.copy_(flat, non_blocking=True)
        return k

    def synchronize(self):
        for st in self.streams:
            st.synchronize()

    def close(self):
        for e in self.engines[1:]:
            e.close()
