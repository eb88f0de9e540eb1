"""ORACLE (test infrastructure, not product code): CPU restatement of the reference's forward graph.

PARITY: PINNED TO THE REFERENCE'S OWN GRAPH CODE, UNPINNED AT THE TENSORFLOW-KERNEL LEVEL.  The reference
(taipingeric/yolo-v4-tf.keras) has no tests, golden vectors or fixtures, and TensorFlow cannot be imported in the build
container.  What pins this restatement: `tests/golden/make_ref_fixtures.py` runs the reference's unmodified
`Yolov4.build_model` -> `yolov4_neck` (+ `load_weights`) from /root/reference with `tensorflow` replaced by a torch-backed
stand-in (`tests/golden/tf_standin.py`), and `tests/test_ref_fixtures.py` holds this module to those outputs (raw heads
within 2e-4) and `yolo4hip.plan` to the traced layer graph.  So layer order, wiring, concat orders, activations, padding,
BN row order and weight layout are the reference's; the per-op arithmetic (Conv2D, BatchNormalization eps 1e-3, pooling
...) follows the documented tf.keras semantics of SURVEY.md section 3.4 on both sides and was never compared with a real
TensorFlow build.

Only `tests/`, `__graft_entry__.smoke()` and `bench.py`'s `cpu_baseline` leg may import this module.

Follows `/root/reference/custom_layers.py`:
  conv            :5-31    Conv2D (+ZeroPadding2D((1,0),(1,0)) & stride 2 'valid' when downsampling,
                           else 'same' stride 1; use_bias = not batch_norm) -> BatchNormalization
                           (Keras default eps 1e-3, inference) -> mish | LeakyReLU(0.1) | nothing
  residual_block  :34-44
  csp_block       :47-69   (route conv is created BEFORE the main-in conv)
  cspdarknet53    :100-138 (convs 0,1 use the default 'leaky'; SPP concat [mp13, mp9, mp5, x])
  yolov4_neck     :141-198
Weights are consumed in Keras creation order == Darknet order (`utils.py:19-21`): every `conv()` call
takes the next entry of the weight list, exactly as layer names conv2d, conv2d_1, ... are assigned.

Tensors are NCHW torch tensors in channels_last memory format (oneDNN's fast path); inputs/outputs of
`yolo_model_forward` are NHWC numpy arrays like Keras `Model.predict`.
"""
import numpy as np
import torch
import torch.nn.functional as F

BN_EPS = 1e-3


def _round_storage(t, storage):
    """Round a float32 tensor to the 16-bit storage type and back (round-to-nearest-even): what a 16-bit pipeline keeps of
    every activation tensor."""
    if storage is None:
        return t
    return t.to(torch.bfloat16 if storage == "bf16" else torch.float16).to(torch.float32)


class _Net:
    def __init__(self, weights, dtype, collect=None, storage=None, trunk=None):
        self.weights = weights
        self.dtype = dtype
        self.i = 0
        self.collect = set(collect or ())
        self.taps = {}
        # storage = 'bf16' | 'f16': EMULATE the 16-bit pipeline of the HIP path on the CPU -- weights and every stored
        # activation tensor rounded to 16 bits, fp32 accumulate / BN / activation, the residual Add inside the 3x3 conv's
        # epilogue (one rounding of the sum), raw heads kept in float32 -- to separate what 16-bit STORAGE costs (this
        # emulation vs the fp32 oracle) from what the kernels add on top (HIP vs this emulation).  Not a reference behaviour.
        self.storage = storage
        # trunk = 'f32' | 'f16' (with storage = 'bf16'): price a WIDER residual trunk -- the tensor `x` of residual_block
        # (custom_layers.py:34-44) carried from Add to Add in float32 / float16 beside its 16-bit rounding, which is what the
        # convs keep consuming.  A diagnostic of what the 23 Adds' roundings cost, not a reference behaviour.
        self.trunk = trunk
        self._wide = None
        self._defer_round = False

    # custom_layers.py:5-31
    def conv(self, x, filters, kernel_size, downsampling=False, activation="leaky", batch_norm=True):
        cw = self.weights[self.i]
        idx = self.i
        self.i += 1
        w = _round_storage(torch.from_numpy(np.ascontiguousarray(cw.w)).to(self.dtype), self.storage)   # OIHW == torch layout
        assert w.shape[0] == filters and w.shape[2] == kernel_size, (idx, w.shape, filters, kernel_size)
        assert w.shape[1] == x.shape[1], (idx, w.shape, x.shape)
        assert (cw.bn is not None) == batch_norm, idx
        if downsampling:
            x = F.pad(x, (1, 0, 1, 0))            # left, right, top, bottom: top & left only (:9-12)
            y = F.conv2d(x, w, None, stride=2, padding=0)
        else:
            y = F.conv2d(x, w, None, stride=1, padding=kernel_size // 2)
        if batch_norm:
            beta, gamma, mean, var = (torch.from_numpy(np.ascontiguousarray(r)).to(self.dtype) for r in cw.bn)
            scale = gamma * torch.rsqrt(var + BN_EPS)
            shift = beta - mean * scale
            y = y * scale.view(1, -1, 1, 1) + shift.view(1, -1, 1, 1)
        else:
            y = y + torch.from_numpy(np.ascontiguousarray(cw.bias)).to(self.dtype).view(1, -1, 1, 1)
        if activation == "mish":
            y = y * torch.tanh(F.softplus(y))     # :6-7
        elif activation == "leaky":
            y = F.leaky_relu(y, 0.1)              # :29-30
        if batch_norm and not self._defer_round:      # (the three head convs are stored in float32)
            y = _round_storage(y, self.storage)
        if idx in self.collect:
            self.taps[idx] = y
        return y

    # custom_layers.py:34-44
    def residual_block(self, x, filters1, filters2, activation="leaky"):
        y = self.conv(x, filters1, 1, activation=activation)
        self._defer_round = True                      # storage emulation: the sum is rounded, not the branch
        y = self.conv(y, filters2, 3, activation=activation)
        self._defer_round = False
        if self.trunk is not None and self.storage is not None:
            wide = (self._wide if self._wide is not None else x) + y
            if self.trunk == "f16":
                wide = wide.to(torch.float16).to(torch.float32)
            self._wide = wide
            out = _round_storage(wide, self.storage)
        else:
            out = _round_storage(x + y, self.storage)
        if (self.i - 1) in self.collect:
            self.taps[("add", self.i - 1)] = out      # what the HIP path stores for this conv (Add fused)
        return out

    # custom_layers.py:47-69
    def csp_block(self, x, residual_out, repeat, residual_bottleneck=False):
        route = self.conv(x, residual_out, 1, activation="mish")
        x = self.conv(x, residual_out, 1, activation="mish")
        self._wide = None                             # (a wider trunk starts at the stage's main-in conv output, as stored)
        for _ in range(repeat):
            x = self.residual_block(x, residual_out // 2 if residual_bottleneck else residual_out,
                                    residual_out, activation="mish")
        self._wide = None
        x = self.conv(x, residual_out, 1, activation="mish")
        return torch.cat([x, route], dim=1)

    # custom_layers.py:100-138
    def cspdarknet53(self, x):
        x = self.conv(x, 32, 3)
        x = self.conv(x, 64, 3, downsampling=True)
        x = self.csp_block(x, 64, 1, residual_bottleneck=True)
        x = self.conv(x, 64, 1, activation="mish")
        x = self.conv(x, 128, 3, activation="mish", downsampling=True)
        x = self.csp_block(x, 64, 2)
        x = self.conv(x, 128, 1, activation="mish")
        x = self.conv(x, 256, 3, activation="mish", downsampling=True)
        x = self.csp_block(x, 128, 8)
        x = self.conv(x, 256, 1, activation="mish")
        route0 = x
        x = self.conv(x, 512, 3, activation="mish", downsampling=True)
        x = self.csp_block(x, 256, 8)
        x = self.conv(x, 512, 1, activation="mish")
        route1 = x
        x = self.conv(x, 1024, 3, activation="mish", downsampling=True)
        x = self.csp_block(x, 512, 4)
        x = self.conv(x, 1024, 1, activation="mish")
        x = self.conv(x, 512, 1)
        x = self.conv(x, 1024, 3)
        x = self.conv(x, 512, 1)
        # MaxPooling2D(pool_size=k, strides=1, padding='same'): windows clipped at the border (:130-134)
        x = torch.cat([F.max_pool2d(x, 13, 1, 6), F.max_pool2d(x, 9, 1, 4), F.max_pool2d(x, 5, 1, 2), x], dim=1)
        x = self.conv(x, 512, 1)
        x = self.conv(x, 1024, 3)
        route2 = self.conv(x, 512, 1)
        return route0, route1, route2

    # custom_layers.py:141-198
    def yolov4_neck(self, x, num_classes):
        route0, route1, route2 = self.cspdarknet53(x)
        route_input = route2
        x = self.conv(route2, 256, 1)
        x = F.interpolate(x, scale_factor=2, mode="nearest")     # UpSampling2D() default: nearest x2
        route1 = self.conv(route1, 256, 1)
        x = torch.cat([route1, x], dim=1)
        x = self.conv(x, 256, 1)
        x = self.conv(x, 512, 3)
        x = self.conv(x, 256, 1)
        x = self.conv(x, 512, 3)
        x = self.conv(x, 256, 1)
        route1 = x
        x = self.conv(x, 128, 1)
        x = F.interpolate(x, scale_factor=2, mode="nearest")
        route0 = self.conv(route0, 128, 1)
        x = torch.cat([route0, x], dim=1)
        x = self.conv(x, 128, 1)
        x = self.conv(x, 256, 3)
        x = self.conv(x, 128, 1)
        x = self.conv(x, 256, 3)
        x = self.conv(x, 128, 1)
        route0 = x
        x = self.conv(x, 256, 3)
        conv_sbbox = self.conv(x, 3 * (num_classes + 5), 1, activation=None, batch_norm=False)
        x = self.conv(route0, 256, 3, downsampling=True)
        x = torch.cat([x, route1], dim=1)
        x = self.conv(x, 256, 1)
        x = self.conv(x, 512, 3)
        x = self.conv(x, 256, 1)
        x = self.conv(x, 512, 3)
        x = self.conv(x, 256, 1)
        route1 = x
        x = self.conv(x, 512, 3)
        conv_mbbox = self.conv(x, 3 * (num_classes + 5), 1, activation=None, batch_norm=False)
        x = self.conv(route1, 512, 3, downsampling=True)
        x = torch.cat([x, route_input], dim=1)
        x = self.conv(x, 512, 1)
        x = self.conv(x, 1024, 3)
        x = self.conv(x, 512, 1)
        x = self.conv(x, 1024, 3)
        x = self.conv(x, 512, 1)
        x = self.conv(x, 1024, 3)
        conv_lbbox = self.conv(x, 3 * (num_classes + 5), 1, activation=None, batch_norm=False)
        return [conv_sbbox, conv_mbbox, conv_lbbox]


def yolo_model_forward(imgs_nhwc, weights, num_classes, dtype=torch.float32, collect=None, threads=None, storage=None, trunk=None):
    """`yolo_model.predict(imgs)` (`models.py:50-52`): NHWC float images in [0,1] -> list of 3 NHWC
    arrays [N,H/8,W/8,3(C+5)], [N,H/16,..], [N,H/32,..] (raw logits).  With `collect=[conv idx...]`
    also returns {idx: NHWC array of that conv's post-activation output}.  `storage` ('bf16' | 'f16'): emulate the 16-bit
    storage pipeline (see _Net) -- a diagnostic, not a reference behaviour."""
    if threads:
        torch.set_num_threads(int(threads))
    x = torch.from_numpy(np.ascontiguousarray(imgs_nhwc)).to(dtype)      # Keras casts to float32
    x = _round_storage(x, storage)                      # the stem's MFMA operand is 16-bit
    x = x.permute(0, 3, 1, 2).contiguous(memory_format=torch.channels_last)
    net = _Net(weights, dtype, collect, storage, trunk)
    with torch.no_grad():
        outs = net.yolov4_neck(x, num_classes)
    assert net.i == len(weights) == 110, net.i
    res = [o.permute(0, 2, 3, 1).contiguous().numpy() for o in outs]
    if collect is not None:
        # int key: that conv's own output; ('add', idx): output of the residual Add that follows conv idx
        taps = {k: v.permute(0, 2, 3, 1).contiguous().numpy() for k, v in net.taps.items()}
        return res, taps
    return res


def conv_block(x_nhwc, cw, k, stride, act, residual_nhwc=None, dtype=torch.float32):
    """One `conv()` unit (+ optional `Add` with `residual`, custom_layers.py:44) on NHWC numpy input;
    used by per-kernel parity tests.  act in {'mish','leaky',None}."""
    net = _Net([cw], dtype)
    x = torch.from_numpy(np.ascontiguousarray(x_nhwc)).to(dtype).permute(0, 3, 1, 2)
    with torch.no_grad():
        y = net.conv(x, cw.w.shape[0], k, downsampling=(stride == 2), activation=act,
                     batch_norm=cw.bn is not None)
        if residual_nhwc is not None:
            y = y + torch.from_numpy(np.ascontiguousarray(residual_nhwc)).to(dtype).permute(0, 3, 1, 2)
    return y.permute(0, 2, 3, 1).contiguous().numpy()


def spp_concat(x_nhwc, dtype=torch.float32):
    """`Concatenate([maxpool13(x), maxpool9(x), maxpool5(x), x])` (custom_layers.py:130-134), NHWC."""
    x = torch.from_numpy(np.ascontiguousarray(x_nhwc)).to(dtype).permute(0, 3, 1, 2)
    y = torch.cat([F.max_pool2d(x, 13, 1, 6), F.max_pool2d(x, 9, 1, 4), F.max_pool2d(x, 5, 1, 2), x], dim=1)
    return y.permute(0, 2, 3, 1).contiguous().numpy()
