"""Darknet `.weights` I/O and the deterministic synthetic weight set.

File format (what the reference's `load_weights` consumes, `utils.py:12-53`): a 20-byte header of five
int32 (`utils.py:16`), then for conv 0..109 in order: for a batch-normalised conv `4*cout` float32 as
rows `[beta, gamma, mean, var]` (`utils.py:28-31`; Keras wants `[gamma, beta, mean, var]`, hence the
reference's row permutation), for the three head convs 93/101/109 `cout` biases (`utils.py:36`); then
`cout*cin*k*k` float32 in Darknet `(out, in, h, w)` order (`utils.py:40-42`).

`flatten()` yields exactly that float stream (without the header): it is the blob the C ABI's
`y4_pack_weights` takes, so a real `yolov4.weights` file and the synthetic set go through one path.

The synthetic set exists because there is no network and no `yolov4.weights` on the build or bench
machines, and the Keras initialiser (`custom_layers.py:22`, N(0, 0.01) with identity BN) collapses every
head logit to ~0.  Recipe: per-layer seeded normal kernels whose std is chosen from an
analytic second-moment walk over the graph so each pre-activation has ~unit second moment; BN gamma 1,
beta ~ N(0, 0.1), mean 0, var 1; head rows scaled/biased per field so that O(10^2..10^3) boxes per image
pass the 0.3 score threshold and NMS has real work.  It is data independent, so every rank of a
multi-GPU job (and the CPU oracle in the tests) regenerates bit-identical weights from the seed.
"""
import math
import os
from dataclasses import dataclass
from typing import List, Optional

import numpy as np

from .plan import ACT_LEAKY, ACT_LINEAR, ACT_MISH, Plan

BN_EPS = 1e-3   # Keras BatchNormalization default, custom_layers.py:26 (NOT Darknet's 1e-6)


@dataclass
class ConvWeights:
    w: np.ndarray                      # float32 (cout, cin, k, k)  -- Darknet OIHW
    bn: Optional[np.ndarray] = None    # float32 (4, cout) rows [beta, gamma, mean, var] (Darknet order)
    bias: Optional[np.ndarray] = None  # float32 (cout,) for the head convs

    def scale_shift(self):
        """Per-channel fp32 `y = conv*scale + shift` equivalent of BN inference / bias."""
        if self.bn is not None:
            beta, gamma, mean, var = self.bn.astype(np.float32)
            scale = (gamma / np.sqrt(var + np.float32(BN_EPS))).astype(np.float32)
            shift = (beta - mean * scale).astype(np.float32)
        else:
            scale = np.ones_like(self.bias, dtype=np.float32)
            shift = self.bias.astype(np.float32)
        return scale, shift


WeightSet = List[ConvWeights]


def n_floats(plan: Plan) -> int:
    return plan.n_params


def flatten(ws: WeightSet) -> np.ndarray:
    parts = []
    for cw in ws:
        parts.append((cw.bn if cw.bn is not None else cw.bias).reshape(-1))
        parts.append(cw.w.reshape(-1))
    return np.ascontiguousarray(np.concatenate(parts), dtype=np.float32)


def unflatten(plan: Plan, flat: np.ndarray) -> WeightSet:
    flat = np.asarray(flat, dtype=np.float32).reshape(-1)
    if flat.size < plan.n_params:
        raise ValueError(f"weight blob too short: {flat.size} floats, plan needs {plan.n_params}")
    ws, off = [], 0
    for c in plan.convs:
        if c.bn:
            bn = flat[off:off + 4 * c.cout].reshape(4, c.cout).copy(); off += 4 * c.cout
            bias = None
        else:
            bias = flat[off:off + c.cout].copy(); off += c.cout
            bn = None
        n = c.n_weights
        w = flat[off:off + n].reshape(c.cout, c.cin, c.k, c.k).copy(); off += n
        ws.append(ConvWeights(w=w, bn=bn, bias=bias))
    return ws


def read_darknet(path: str, plan: Plan):
    """Returns (WeightSet, header int32[5], n_unread_floats).  The reference prints 'all weights read'
    or a (buggy, always 0) unread count (`utils.py:50-53`); we return the true count."""
    with open(path, "rb") as f:
        header = np.fromfile(f, dtype=np.int32, count=5)
        if header.size != 5:
            raise ValueError(f"{path}: truncated header")
        flat = np.fromfile(f, dtype=np.float32)
    ws = unflatten(plan, flat)
    return ws, header, int(flat.size - plan.n_params)


def write_darknet(path: str, ws: WeightSet, header=(0, 2, 5, 0, 0)):
    with open(path, "wb") as f:
        np.asarray(header, dtype=np.int32).tofile(f)
        flatten(ws).tofile(f)


# ----------------------------------------------------------------------------- synthetic weights
def _gauss_moments(fn, mu=0.0):
    z = np.linspace(-8.0, 8.0, 32001)
    pdf = np.exp(-0.5 * z * z) / math.sqrt(2 * math.pi)
    y = fn(z + mu)
    dz = z[1] - z[0]
    return float((y * pdf).sum() * dz), float((y * y * pdf).sum() * dz)


def _mish(z):
    return z * np.tanh(np.log1p(np.exp(z)))


def _leaky(z):
    return np.where(z > 0, z, 0.1 * z)


# crude model of a stride-1 'same' max-pool (spatially correlated inputs): mean shifts up by k std
_POOL_SHIFT = {5: 1.5, 9: 2.0, 13: 2.3}

RESIDUAL_GAMMA = 0.25
SYNTH_BETA_MEAN = 1.0     # BN beta mean: keeps most pre-activations in the near-linear region -> well-conditioned net
HEAD_GAIN = 1.5          # std of class / objectness logits
HEAD_WH_GAIN = 0.35      # std of tw, th  (exp(0.35 z): boxes within ~0.4x..2.5x of the anchor)
HEAD_OBJ_BIAS = -2.8


def head_cls_bias(num_classes: int) -> float:
    """Class-logit bias: keeps ~1-2 classes per confident box whatever the class count."""
    return -0.5 - 0.6 * math.log(num_classes)


def synth_weights(plan: Plan, seed: int = 0) -> WeightSet:
    mean = {"input": 0.5}
    m2 = {"input": 1.0 / 3.0}     # U[0,1) input
    ws: List[Optional[ConvWeights]] = [None] * len(plan.convs)
    # convs whose output is the branch operand of a residual Add get a small BN gamma (the usual
    # "zero-init residual" practice): with gamma 1 each of the 23 blocks x + f(x) amplifies any
    # perturbation ~1.4x, which makes a RANDOM network chaotic (fp32 rounding noise grows ~10^3 through
    # the stack, for the CPU oracle and the HIP path alike) -- unlike a trained one.
    moments = {ACT_MISH: _gauss_moments(_mish, SYNTH_BETA_MEAN), ACT_LEAKY: _gauss_moments(_leaky, SYNTH_BETA_MEAN),
               ACT_LINEAR: (0.0, 1.0)}
    branch_convs = {int(op.srcs[1][1:]) for op in plan.ops if op.kind == "add"}
    for op in plan.ops:
        if op.kind == "conv":
            c = plan.convs[op.conv]
            src = op.srcs[0]
            rng = np.random.default_rng([seed, c.idx])
            fan_in = c.k * c.k * c.cin
            # zero-sum rows remove the common-mode response to the (positive) input mean -- what BN's
            # mean subtraction does after training -- so only the input's variance drives the output
            std = 1.0 / math.sqrt(fan_in * max(m2[src] - mean[src] ** 2, 1e-3))
            w = rng.standard_normal((c.cout, c.cin, c.k, c.k), dtype=np.float32)
            w -= w.mean(axis=(1, 2, 3), keepdims=True, dtype=np.float32)
            if c.bn:
                w *= np.float32(std)
                bn = np.zeros((4, c.cout), np.float32)
                bn[0] = rng.standard_normal(c.cout, dtype=np.float32) * np.float32(0.1) + np.float32(SYNTH_BETA_MEAN)  # beta
                bn[1] = RESIDUAL_GAMMA if c.idx in branch_convs else 1.0                  # gamma
                bn[2] = 0.0                                                               # mean
                bn[3] = 1.0                                                               # var
                ws[c.idx] = ConvWeights(w=w, bn=bn)
            else:
                nf = 5 + plan.num_classes
                field_of = np.arange(c.cout) % nf
                gain = np.where(field_of < 2, HEAD_GAIN, np.where(field_of < 4, HEAD_WH_GAIN, HEAD_GAIN))
                w *= (gain * std).astype(np.float32)[:, None, None, None]
                bias = np.where(field_of == 4, HEAD_OBJ_BIAS,
                                np.where(field_of > 4, head_cls_bias(plan.num_classes), 0.0)).astype(np.float32)
                bias += rng.standard_normal(c.cout, dtype=np.float32) * np.float32(0.05)
                ws[c.idx] = ConvWeights(w=w, bias=bias.astype(np.float32))
            mean[op.dst], m2[op.dst] = moments[c.act]
            if c.idx in branch_convs:
                mean[op.dst] *= RESIDUAL_GAMMA
                m2[op.dst] *= RESIDUAL_GAMMA ** 2
        elif op.kind == "add":
            a, b = op.srcs
            mean[op.dst] = mean[a] + mean[b]
            m2[op.dst] = m2[a] + m2[b] + 2.0 * mean[a] * mean[b]
        elif op.kind == "concat":
            tot = sum(plan.chans[s] for s in op.srcs)
            mean[op.dst] = sum(mean[s] * plan.chans[s] for s in op.srcs) / tot
            m2[op.dst] = sum(m2[s] * plan.chans[s] for s in op.srcs) / tot
        elif op.kind == "maxpool":
            s = op.srcs[0]
            std_s = math.sqrt(max(m2[s] - mean[s] ** 2, 1e-6))
            mean[op.dst] = mean[s] + _POOL_SHIFT.get(op.k, 2.0) * std_s
            m2[op.dst] = mean[op.dst] ** 2 + 0.5 * std_s ** 2
        elif op.kind == "upsample":
            s = op.srcs[0]
            mean[op.dst], m2[op.dst] = mean[s], m2[s]
        else:
            raise ValueError(op.kind)
    return ws  # type: ignore[return-value]


def synth_images(n: int, img_size: int, seed: int = 0, first_index: int = 0) -> np.ndarray:
    """U[0,1) float32 [n, H, W, 3]; the stream of image i depends only on (seed, first_index + i), so a
    rank's shard of a multi-GPU batch is bit-identical to the same rows of the single-GPU batch."""
    out = np.empty((n, img_size, img_size, 3), np.float32)
    for i in range(n):
        rng = np.random.default_rng([seed, 0x1A6E, first_index + i])
        out[i] = rng.random((img_size, img_size, 3), dtype=np.float32)
    return out


def weights_path_kind(path: Optional[str]) -> str:
    if not path:
        return "none"
    ext = os.path.splitext(path)[1].lower()
    return "darknet" if ext == ".weights" else "unknown"
