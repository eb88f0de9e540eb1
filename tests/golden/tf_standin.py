"""A torch-backed stand-in for the slice of `tensorflow` / `tf.keras` that the reference's inference path touches.

BUILD-CONTAINER TOOLING for `make_ref_fixtures.py`, nothing else: TensorFlow cannot be installed here, so to run the
reference's OWN Python (`/root/reference/custom_layers.py`, `models.py`, `utils.py`) -- its graph construction, layer
creation order, concat orders, decode formulas, NMS wrapper, weight loader -- this module is registered as
`sys.modules['tensorflow']` before the reference is imported.  What is the reference's and what is this file's:

  the reference's (executed unmodified from /root/reference): which layers are created in which order with which
      arguments, how tensors are wired (`conv`, `residual_block`, `csp_block`, `cspdarknet53`, `yolov4_neck`),
      `get_boxes` / `yolov4_head` arithmetic expressions, `nms`'s flatten / concat / `obj*cls` / normalisation, the call
      of `combined_non_max_suppression` with its arguments, `load_weights`' file walk, row permutation and transposes,
      `Yolov4.__init__ / build_model / predict_img / export_* / eval_map`, `get_detection_data`, `voc_ap`.
  this file's (the op ARITHMETIC is a stand-in, float32 on torch-CPU, written from the documented Keras / TF semantics):
      Conv2D (cross-correlation, HWIO kernel, 'same' = (k-1)/2 both sides at stride 1, 'valid' = none),
      BatchNormalization inference (epsilon 1e-3 default), LeakyReLU, ZeroPadding2D, Add, Concatenate, MaxPooling2D
      ('same' = windows clipped at the border), UpSampling2D (nearest), the `tf.*` element-wise / shape functions, and
      `tf.image.combined_non_max_suppression` (per-class greedy NMS, strict `>` thresholds, merge by score, clip).

Graph mode: tensors are lazy nodes (`T`) built by calling layers on `layers.Input`; `Model.predict(x)` evaluates them.
With concrete inputs (numpy arrays, as in `Yolov4.predict_nonms`) every function computes at once, like eager TF.
Layers get Keras' automatic names (`conv2d`, `conv2d_1`, ..., reset by `backend.clear_session()`), which
`utils.load_weights` relies on (`utils.py:20-21`), and every layer call is appended to `TRACE`.
"""
import re
import sys
import types

import numpy as np
import torch
import torch.nn.functional as F

sys.setrecursionlimit(20000)

TRACE = []            # one record per layer call / tf function call, in creation order
_UIDS = {}            # Keras-style name counters
_LAYERS = {}          # name -> layer (since the last clear_session)
float32, int32 = torch.float32, torch.int32


class T:
    """A tensor: concrete (`value` set) or a lazy graph node (`fn`, `parents`)."""
    _count = 0

    def __init__(self, value=None, fn=None, parents=(), shape=None, op="const"):
        self.value, self.fn, self.parents, self.op = value, fn, parents, op
        self.shape = tuple(value.shape) if value is not None and hasattr(value, "shape") else shape
        T._count += 1
        self.uid = T._count

    @property
    def symbolic(self):
        return self.value is None

    def numpy(self):
        assert not self.symbolic, "symbolic tensor has no value"
        return self.value.numpy() if isinstance(self.value, torch.Tensor) else np.asarray(self.value)

    def __getitem__(self, idx):
        return _op("getitem", lambda v: v[idx], self)

    # element-wise: the static shape (only layers need it, for their channel count) is the tensor operand's
    def __mul__(self, o): return _op("mul", _binary(torch.mul), self, o, shape=self.shape)
    def __rmul__(self, o): return _op("mul", _binary(torch.mul), o, self, shape=self.shape)
    def __add__(self, o): return _op("add_", _binary(torch.add), self, o, shape=self.shape)
    def __radd__(self, o): return _op("add_", _binary(torch.add), o, self, shape=self.shape)
    def __sub__(self, o): return _op("sub", _binary(torch.sub), self, o, shape=self.shape)
    def __rsub__(self, o): return _op("sub", _binary(torch.sub), o, self, shape=self.shape)
    def __truediv__(self, o): return _op("div", _binary(torch.div), self, o, shape=self.shape)


def _binary(fn):
    def run(a, b):
        # tf.convert_to_tensor semantics: a Python / numpy operand takes the dtype of the tensor operand
        ref = a if isinstance(a, torch.Tensor) else b
        a = a if isinstance(a, torch.Tensor) else torch.as_tensor(np.asarray(a)).to(ref.dtype)
        b = b if isinstance(b, torch.Tensor) else torch.as_tensor(np.asarray(b)).to(ref.dtype)
        return fn(a, b)
    return run


def _walk(x, fn):
    if isinstance(x, (list, tuple)):
        return type(x)(_walk(v, fn) for v in x)
    return fn(x)


def _find(x, out):
    if isinstance(x, (list, tuple)):
        for v in x:
            _find(v, out)
    elif isinstance(x, T):
        out.append(x)
    return out


def _concrete(x):
    """numpy / T(concrete) -> torch; 0-d integer tensors become Python ints (they are used as shapes)."""
    if isinstance(x, T):
        assert not x.symbolic
        x = x.value
    if isinstance(x, np.ndarray):
        x = torch.from_numpy(np.ascontiguousarray(x))
        if x.dtype == torch.float64:
            x = x.float()
    if isinstance(x, torch.Tensor) and x.dim() == 0 and not x.dtype.is_floating_point:
        return int(x)
    return x


def _op(name, fn, *args, n_out=None, shape=None, record=True):
    """Apply `fn` to args (T's anywhere inside nested lists are resolved).  Lazy if any T is symbolic."""
    ts = _find(list(args), [])
    if any(t.symbolic for t in ts):
        def run(memo):
            return fn(*_walk(list(args), lambda v: _concrete(_eval(v, memo) if isinstance(v, T) else v)))
        node = T(fn=run, parents=tuple(ts), shape=shape, op=name)
        if n_out is None:
            return node
        return [T(fn=(lambda memo, i=i: _eval(node, memo)[i]), parents=(node,), op=f"{name}[{i}]") for i in range(n_out)]
    res = fn(*_walk(list(args), _concrete))
    if n_out is None:
        return T(value=res, op=name)
    return [T(value=r, op=f"{name}[{i}]") for i, r in enumerate(res)]


def _eval(t, memo):
    if not t.symbolic:
        return t.value
    if t.uid not in memo:
        memo[t.uid] = t.fn(memo)
    return memo[t.uid]


# ------------------------------------------------------------------------------------------------ keras layers
def _snake(cls_name):
    s = re.sub(r"(.)([A-Z][a-z0-9]+)", r"\1_\2", cls_name)          # Keras' to_snake_case
    return re.sub(r"([a-z])([A-Z])", r"\1_\2", s).lower()


class Layer:
    def __init__(self, name=None, **_):
        base = _snake(type(self).__name__)
        if name is None:
            n = _UIDS.get(base, 0)
            _UIDS[base] = n + 1
            name = base if n == 0 else f"{base}_{n}"
        self.name = name
        _LAYERS[name] = self
        self.input_shape = None

    def out_shape(self, shp):
        return shp

    def config(self):
        return {}

    def __call__(self, x):
        ins = list(x) if isinstance(x, (list, tuple)) else [x]
        shapes = [t.shape for t in ins]
        self.input_shape = shapes if isinstance(x, (list, tuple)) else shapes[0]
        self.build(self.input_shape)
        oshape = self.out_shape(self.input_shape)
        out = _op(self.name, lambda *v: self.compute(list(v) if isinstance(x, (list, tuple)) else v[0]), *ins, shape=oshape)
        out.layer = self
        TRACE.append({"kind": "layer", "type": type(self).__name__, "name": self.name, "config": self.config(),
                      "inputs": [t.uid for t in ins], "output": out.uid,
                      "in_shapes": [list(s) if s else None for s in shapes], "out_shape": list(oshape) if oshape else None})
        return out

    def build(self, shp):
        pass


class Conv2D(Layer):
    def __init__(self, filters, kernel_size, strides=1, padding="valid", use_bias=True, kernel_initializer=None, **kw):
        super().__init__(**kw)
        self.filters = int(filters)
        self.kernel_size = (kernel_size, kernel_size) if isinstance(kernel_size, int) else tuple(kernel_size)
        self.strides = (strides, strides) if isinstance(strides, int) else tuple(strides)
        self.padding, self.use_bias, self.kernel_initializer = padding, use_bias, kernel_initializer
        self.kernel = self.bias = None

    def build(self, shp):
        if self.kernel is None:
            kshape = (*self.kernel_size, shp[-1], self.filters)                       # HWIO
            init = self.kernel_initializer
            g = torch.Generator().manual_seed(len(_LAYERS))
            std = getattr(init, "stddev", 0.05)
            self.kernel = torch.randn(kshape, generator=g) * std + getattr(init, "mean", 0.0)
            self.bias = torch.zeros(self.filters) if self.use_bias else None

    def out_shape(self, shp):
        n, h, w, _ = shp
        if self.padding == "same":
            assert self.strides == (1, 1), "stand-in: 'same' only at stride 1 (all the reference uses)"
            return (n, h, w, self.filters)
        k, s = self.kernel_size[0], self.strides[0]
        return (n, (h - k) // s + 1, (w - k) // s + 1, self.filters)

    def config(self):
        return {"filters": self.filters, "kernel_size": list(self.kernel_size), "strides": list(self.strides),
                "padding": self.padding, "use_bias": self.use_bias}

    def set_weights(self, ws):
        ws = [torch.from_numpy(np.ascontiguousarray(np.asarray(w, dtype=np.float32))) for w in ws]
        assert tuple(ws[0].shape) == tuple(self.kernel.shape), (self.name, ws[0].shape, self.kernel.shape)
        assert len(ws) == (2 if self.use_bias else 1), self.name
        self.kernel = ws[0]
        if self.use_bias:
            assert tuple(ws[1].shape) == (self.filters,)
            self.bias = ws[1]

    def get_weights(self):
        return [self.kernel.numpy()] + ([self.bias.numpy()] if self.use_bias else [])

    def compute(self, x):
        w = self.kernel.permute(3, 2, 0, 1).contiguous()                               # HWIO -> OIHW
        pad = (self.kernel_size[0] - 1) // 2 if self.padding == "same" else 0
        y = F.conv2d(x.permute(0, 3, 1, 2), w, self.bias, stride=self.strides, padding=pad)
        return y.permute(0, 2, 3, 1).contiguous()


class BatchNormalization(Layer):
    def __init__(self, axis=-1, momentum=0.99, epsilon=1e-3, **kw):
        super().__init__(**kw)
        self.epsilon = epsilon
        self.w = None

    def build(self, shp):
        if self.w is None:
            c = shp[-1]
            self.w = [torch.ones(c), torch.zeros(c), torch.zeros(c), torch.ones(c)]   # gamma, beta, moving mean, moving var

    def config(self):
        return {"epsilon": self.epsilon}

    def set_weights(self, ws):
        ws = [torch.from_numpy(np.ascontiguousarray(np.asarray(w, dtype=np.float32))) for w in ws]
        assert len(ws) == 4 and all(tuple(w.shape) == tuple(self.w[0].shape) for w in ws), self.name
        self.w = ws

    def get_weights(self):
        return [w.numpy() for w in self.w]

    def compute(self, x):
        gamma, beta, mean, var = self.w
        return (x - mean) * (gamma * torch.rsqrt(var + self.epsilon)) + beta


class LeakyReLU(Layer):
    def __init__(self, alpha=0.3, **kw):
        super().__init__(**kw)
        self.alpha = alpha

    def config(self):
        return {"alpha": self.alpha}

    def compute(self, x):
        return F.leaky_relu(x, self.alpha)


class ZeroPadding2D(Layer):
    def __init__(self, padding=(1, 1), **kw):
        super().__init__(**kw)
        (self.top, self.bottom), (self.left, self.right) = padding

    def out_shape(self, shp):
        n, h, w, c = shp
        return (n, h + self.top + self.bottom, w + self.left + self.right, c)

    def config(self):
        return {"padding": [[self.top, self.bottom], [self.left, self.right]]}

    def compute(self, x):
        return F.pad(x, (0, 0, self.left, self.right, self.top, self.bottom))


class Add(Layer):
    def out_shape(self, shp):
        return shp[0]

    def compute(self, xs):
        return xs[0] + xs[1]


class Concatenate(Layer):
    def __init__(self, axis=-1, **kw):
        super().__init__(**kw)
        self.axis = axis

    def out_shape(self, shp):
        return (*shp[0][:-1], sum(s[-1] for s in shp))

    def config(self):
        return {"axis": self.axis}

    def compute(self, xs):
        return torch.cat(xs, dim=self.axis)


class MaxPooling2D(Layer):
    def __init__(self, pool_size=2, strides=None, padding="valid", **kw):
        super().__init__(**kw)
        self.pool_size, self.strides, self.padding = pool_size, strides or pool_size, padding

    def config(self):
        return {"pool_size": self.pool_size, "strides": self.strides, "padding": self.padding}

    def compute(self, x):
        assert self.strides == 1 and self.padding == "same" and self.pool_size % 2 == 1, "stand-in: SPP pooling only"
        y = F.max_pool2d(x.permute(0, 3, 1, 2), self.pool_size, 1, self.pool_size // 2)   # pads with -inf: clipped windows
        return y.permute(0, 2, 3, 1).contiguous()


class UpSampling2D(Layer):
    def __init__(self, size=(2, 2), interpolation="nearest", **kw):
        super().__init__(**kw)
        self.size = size

    def out_shape(self, shp):
        n, h, w, c = shp
        return (n, h * self.size[0], w * self.size[1], c)

    def config(self):
        return {"size": list(self.size)}

    def compute(self, x):
        return x.repeat_interleave(self.size[0], dim=1).repeat_interleave(self.size[1], dim=2)


class Lambda(Layer):
    """Only the training loss uses it (`models.py:60-63`): never evaluated here."""

    def __init__(self, function, name=None, arguments=None, **kw):
        super().__init__(name=name, **kw)

    def __call__(self, x):
        def never(memo):
            raise RuntimeError("the training graph is not evaluable under the stand-in")
        return T(fn=never, op=self.name)


def Input(shape=None, name=None, **_):
    n = _UIDS.get("input", 0) + 1
    _UIDS["input"] = n
    t = T(fn=None, shape=(None, *shape), op=name or f"input_{n}")
    t.is_input = True
    TRACE.append({"kind": "input", "name": t.op, "output": t.uid, "out_shape": [None, *shape]})
    return t


class Model:
    def __init__(self, inputs, outputs, name=None):
        self.input, self.output = inputs, outputs
        self.inputs = list(inputs) if isinstance(inputs, (list, tuple)) else [inputs]
        self.outputs = list(outputs) if isinstance(outputs, (list, tuple)) else [outputs]

    def get_layer(self, name):
        return _LAYERS[name]

    def predict(self, x, batch_size=None, **_):
        xs = x if isinstance(x, (list, tuple)) else [x]
        memo = {}
        for t, v in zip(self.inputs, xs):
            memo[t.uid] = torch.from_numpy(np.ascontiguousarray(np.asarray(v, dtype=np.float32)))   # Keras casts to float32
        with torch.no_grad():
            outs = [_eval(t, memo) for t in self.outputs]
        outs = [o.numpy() if isinstance(o, torch.Tensor) else np.asarray(o) for o in outs]
        return outs if isinstance(self.output, (list, tuple)) else outs[0]

    def compile(self, *a, **k):
        pass

    def save(self, path):
        raise NotImplementedError


class RandomNormal:
    def __init__(self, mean=0.0, stddev=0.05, seed=None):
        self.mean, self.stddev = mean, stddev


class Adam:
    def __init__(self, *a, **k):
        pass


# ------------------------------------------------------------------------------------------------ tf.* functions
def _unary(name, fn, keeps_shape=True):
    return lambda x, name_=None: _op(name, fn, x, shape=getattr(x, "shape", None) if keeps_shape else None)


def _reshape(x, shape):
    return _op("reshape", lambda v, s: v.reshape(tuple(int(d) for d in s)), x, list(shape))


def _split(x, sizes, axis=0):
    return _op("split", lambda v: torch.split(v, list(sizes), dim=axis), x, n_out=len(sizes))


def _concat(xs, axis):
    return _op("concat", lambda vs: torch.cat([v if isinstance(v, torch.Tensor) else torch.as_tensor(v) for v in vs], dim=axis), list(xs))


def _meshgrid(a, b):
    # tf.meshgrid default indexing='xy': outputs have shape (len(b), len(a)); the first varies along columns
    return _op("meshgrid", lambda u, v: (u[None, :].expand(len(v), len(u)), v[:, None].expand(len(v), len(u))), a, b, n_out=2)


def _zeros(shape, dtype=float32):
    return _op("zeros", lambda s: torch.zeros(tuple(int(d) for d in s), dtype=dtype), list(shape))


def combined_non_max_suppression(boxes, scores, max_output_size_per_class, max_total_size, iou_threshold=0.5,
                                 score_threshold=float("-inf"), pad_per_class=False, clip_boxes=True, name=None):
    """STAND-IN for TensorFlow's CombinedNonMaxSuppression (documented behaviour): boxes [N,nb,q,4] with q == 1 (shared
    by all classes), scores [N,nb,C].  Per image and class: candidates with score > score_threshold in descending score
    (ties: lower box index first), greedily kept unless IoU with an already kept box of the class is > iou_threshold,
    at most max_output_size_per_class; all classes merged by descending score (ties: box index, then class),
    truncated to max_total_size, zero padded, coordinates clipped to [0,1].  float32 IoU on min/max-normalised corners,
    0 when an area is not positive."""
    TRACE.append({"kind": "tf", "name": "combined_non_max_suppression", "config": {
        "max_output_size_per_class": max_output_size_per_class, "max_total_size": max_total_size,
        "iou_threshold": iou_threshold, "score_threshold": score_threshold, "pad_per_class": pad_per_class,
        "clip_boxes": clip_boxes}})

    def run(bx, sc):
        bx, sc = bx.numpy().astype(np.float32), sc.numpy().astype(np.float32)
        assert bx.ndim == 4 and bx.shape[2] == 1 and not pad_per_class
        n, nb, ncls = sc.shape
        ob = np.zeros((n, max_total_size, 4), np.float32); osc = np.zeros((n, max_total_size), np.float32)
        ocl = np.zeros((n, max_total_size), np.float32); ov = np.zeros((n,), np.int32)
        thr_i, thr_s = np.float32(iou_threshold), np.float32(score_threshold)

        def iou(p, q):
            py0, py1 = min(p[0], p[2]), max(p[0], p[2]); px0, px1 = min(p[1], p[3]), max(p[1], p[3])
            qy0, qy1 = min(q[0], q[2]), max(q[0], q[2]); qx0, qx1 = min(q[1], q[3]), max(q[1], q[3])
            ap, aq = np.float32((py1 - py0) * (px1 - px0)), np.float32((qy1 - qy0) * (qx1 - qx0))
            if ap <= 0 or aq <= 0:
                return np.float32(0)
            ih = max(np.float32(min(py1, qy1) - max(py0, qy0)), np.float32(0))
            iw = max(np.float32(min(px1, qx1) - max(px0, qx0)), np.float32(0))
            inter = np.float32(ih * iw)
            return np.float32(inter / np.float32(ap + aq - inter))

        for b in range(n):
            found = []
            for c in range(ncls):
                idx = [i for i in range(nb) if sc[b, i, c] > thr_s]
                idx.sort(key=lambda i: (-float(sc[b, i, c]), i))
                kept = []
                for i in idx:
                    if len(kept) == max_output_size_per_class:
                        break
                    if all(not (iou(bx[b, i, 0], bx[b, j, 0]) > thr_i) for j in kept):
                        kept.append(i)
                found += [(-float(sc[b, i, c]), i, c) for i in kept]
            found.sort()
            found = found[:max_total_size]
            ov[b] = len(found)
            for k, (ns, i, c) in enumerate(found):
                ob[b, k] = np.clip(bx[b, i, 0], 0, 1) if clip_boxes else bx[b, i, 0]
                osc[b, k], ocl[b, k] = sc[b, i, c], c
        return tuple(torch.from_numpy(a) for a in (ob, osc, ocl, ov))

    return tuple(_op("combined_nms", run, boxes, scores, n_out=4))


def install():
    """Registers the stand-in as `tensorflow` (+ the submodules the reference imports).  Returns the module."""
    tf = types.ModuleType("tensorflow")
    tf.__version__ = "stand-in (tests/golden/tf_standin.py)"
    tf.float32, tf.int32 = float32, int32
    tf.reshape, tf.split, tf.concat, tf.meshgrid, tf.zeros = _reshape, _split, _concat, _meshgrid, _zeros
    tf.sigmoid = _unary("sigmoid", torch.sigmoid)
    tf.exp = _unary("exp", torch.exp)
    tf.shape = _unary("shape", lambda v: torch.tensor(list(v.shape), dtype=torch.int32), keeps_shape=False)
    tf.range = lambda n: _op("range", lambda k: torch.arange(int(k), dtype=torch.int32), n)
    tf.stack = lambda xs, axis=0: _op("stack", lambda vs: torch.stack(list(vs), dim=axis), list(xs))
    tf.expand_dims = lambda x, axis: _op("expand_dims", lambda v: v.unsqueeze(axis), x)
    tf.cast = lambda x, dtype: _op("cast", lambda v: v.to(dtype), x)
    tf.math = types.ModuleType("tensorflow.math")
    tf.math.tanh = _unary("tanh", torch.tanh)
    tf.math.softplus = _unary("softplus", F.softplus)
    tf.image = types.ModuleType("tensorflow.image")
    tf.image.combined_non_max_suppression = combined_non_max_suppression
    keras = types.ModuleType("tensorflow.keras")
    layers = types.ModuleType("tensorflow.keras.layers")
    for cls in (Conv2D, BatchNormalization, LeakyReLU, ZeroPadding2D, Add, Concatenate, MaxPooling2D, UpSampling2D, Lambda):
        setattr(layers, cls.__name__, cls)
    layers.Input = Input
    models = types.ModuleType("tensorflow.keras.models")
    models.Model = Model
    initializers = types.ModuleType("tensorflow.keras.initializers")
    initializers.RandomNormal = RandomNormal
    optimizers = types.ModuleType("tensorflow.keras.optimizers")
    optimizers.Adam = Adam
    backend = types.ModuleType("tensorflow.keras.backend")

    def clear_session():
        _UIDS.clear(); _LAYERS.clear(); TRACE.clear()
    backend.clear_session = clear_session
    utils = types.ModuleType("tensorflow.keras.utils")
    utils.Sequence = object
    callbacks = types.ModuleType("tensorflow.keras.callbacks")
    callbacks.Callback = object
    keras.layers, keras.models, keras.initializers, keras.optimizers = layers, models, initializers, optimizers
    keras.backend, keras.utils, keras.callbacks = backend, utils, callbacks
    tf.keras = keras
    for name, mod in (("tensorflow", tf), ("tensorflow.math", tf.math), ("tensorflow.image", tf.image),
                      ("tensorflow.keras", keras), ("tensorflow.keras.layers", layers), ("tensorflow.keras.models", models),
                      ("tensorflow.keras.initializers", initializers), ("tensorflow.keras.optimizers", optimizers),
                      ("tensorflow.keras.backend", backend), ("tensorflow.keras.utils", utils),
                      ("tensorflow.keras.callbacks", callbacks)):
        sys.modules[name] = mod
    return tf
