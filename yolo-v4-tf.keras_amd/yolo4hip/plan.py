"""The 110-conv YOLOv4 inference plan: (img_size, num_classes) -> layer table + op list.

This is the host-side statement of the graph the reference builds with tf.keras in
`custom_layers.py:100-198` (cspdarknet53 + SPP + PANet neck + 3 raw heads).  Conv index ==
Keras creation order == Darknet blob order (`utils.py:19-21`), so the table doubles as the
weight-file layout.

Reference behaviours reproduced on purpose:
  * convs 0 and 1 are LeakyReLU, not Mish (`custom_layers.py:101-102` use the default activation)
  * in a CSP block the *route* conv is created before the main-in conv (`custom_layers.py:58-60`)
  * in the PANet top-down path the upsample-branch conv is created before the lateral conv
    (`custom_layers.py:146-148`, `:158-160`)
  * concat orders: CSP [x, route] (:68); SPP [mp13, mp9, mp5, x] (:130-134); top-down
    [lateral, upsampled] (:149,:161); bottom-up [downsampled, route] (:174,:187)
Generalised on purpose: grid sizes are img_size // stride instead of the hard-coded 52/26/13
(`custom_layers.py:204,208,212`), which is what lets the 608x608 configs exist at all.

The C++ runtime (csrc/runtime.hip: build_plan) builds the same table independently; tests compare the two through
`y4_layer_info`.
"""
from dataclasses import dataclass, field
from typing import List, Optional, Tuple

HEAD_CONV_IDXS = (93, 101, 109)   # `utils.py:14`
NUM_CONVS = 110                   # `utils.py:13`

ACT_LINEAR, ACT_LEAKY, ACT_MISH = 0, 1, 2
ACT_NAMES = {ACT_LINEAR: "linear", ACT_LEAKY: "leaky", ACT_MISH: "mish"}


@dataclass
class ConvSpec:
    idx: int
    k: int              # kernel size 1 or 3
    s: int              # stride 1 or 2 (2 == ZeroPadding2D((1,0),(1,0)) + 'valid', custom_layers.py:9-12)
    cin: int
    cout: int
    act: int            # ACT_*
    bn: bool            # False only for the 3 head convs (bias instead), custom_layers.py:20
    in_side: int = 0
    out_side: int = 0

    @property
    def flops_per_image(self) -> int:
        return 2 * self.k * self.k * self.cin * self.cout * self.out_side * self.out_side

    @property
    def n_weights(self) -> int:
        return self.k * self.k * self.cin * self.cout


@dataclass
class Op:
    kind: str                       # 'conv' | 'add' | 'concat' | 'maxpool' | 'upsample'
    dst: str
    srcs: Tuple[str, ...]
    conv: Optional[int] = None      # conv index for kind == 'conv'
    k: int = 0                      # pool size for 'maxpool'


@dataclass
class Plan:
    img_size: int
    num_classes: int
    convs: List[ConvSpec] = field(default_factory=list)
    ops: List[Op] = field(default_factory=list)
    sides: dict = field(default_factory=dict)      # tensor name -> spatial side
    chans: dict = field(default_factory=dict)      # tensor name -> channels
    heads: Tuple[str, str, str] = ("", "", "")     # conv_sbbox, conv_mbbox, conv_lbbox tensor names

    @property
    def strides(self):
        return (8, 16, 32)

    @property
    def grids(self):
        return tuple(self.img_size // s for s in self.strides)

    @property
    def num_boxes(self) -> int:
        return sum(3 * g * g for g in self.grids)

    @property
    def flops_per_image(self) -> int:
        return sum(c.flops_per_image for c in self.convs)

    @property
    def n_params(self) -> int:
        return sum(c.n_weights + (4 * c.cout if c.bn else c.cout) for c in self.convs)


class _Builder:
    def __init__(self, img_size: int, num_classes: int):
        self.p = Plan(img_size=img_size, num_classes=num_classes)
        self.p.sides["input"] = img_size
        self.p.chans["input"] = 3
        self._tmp = 0

    def _name(self, stem):
        self._tmp += 1
        return f"{stem}{self._tmp}"

    def conv(self, x, filters, k, down=False, act=ACT_LEAKY, bn=True):
        p = self.p
        idx = len(p.convs)
        side = p.sides[x]
        out_side = side // 2 if down else side
        p.convs.append(ConvSpec(idx, k, 2 if down else 1, p.chans[x], filters, act, bn, side, out_side))
        dst = f"c{idx}"
        p.ops.append(Op("conv", dst, (x,), conv=idx))
        p.sides[dst], p.chans[dst] = out_side, filters
        return dst

    def add(self, a, b):
        dst = self._name("add")
        self.p.ops.append(Op("add", dst, (a, b)))
        self.p.sides[dst], self.p.chans[dst] = self.p.sides[a], self.p.chans[a]
        return dst

    def concat(self, *xs):
        dst = self._name("cat")
        self.p.ops.append(Op("concat", dst, tuple(xs)))
        self.p.sides[dst] = self.p.sides[xs[0]]
        self.p.chans[dst] = sum(self.p.chans[x] for x in xs)
        return dst

    def maxpool(self, x, k):
        dst = self._name(f"mp{k}_")
        self.p.ops.append(Op("maxpool", dst, (x,), k=k))
        self.p.sides[dst], self.p.chans[dst] = self.p.sides[x], self.p.chans[x]
        return dst

    def upsample(self, x):
        dst = self._name("up")
        self.p.ops.append(Op("upsample", dst, (x,)))
        self.p.sides[dst], self.p.chans[dst] = 2 * self.p.sides[x], self.p.chans[x]
        return dst

    # custom_layers.py:34-44
    def residual(self, x, f1, f2, act):
        y = self.conv(x, f1, 1, act=act)
        y = self.conv(y, f2, 3, act=act)
        return self.add(x, y)

    # custom_layers.py:47-69
    def csp(self, x, width, repeat, bottleneck=False):
        route = self.conv(x, width, 1, act=ACT_MISH)
        x = self.conv(x, width, 1, act=ACT_MISH)
        for _ in range(repeat):
            x = self.residual(x, width // 2 if bottleneck else width, width, ACT_MISH)
        x = self.conv(x, width, 1, act=ACT_MISH)
        return self.concat(x, route)


def build_plan(img_size: int = 416, num_classes: int = 80) -> Plan:
    """Build the plan for a square `img_size` (multiple of 32, `models.py:23-24`)."""
    assert img_size % 32 == 0, "must be a multiple of last stride"
    assert num_classes > 0, "no classes detected!"
    b = _Builder(img_size, num_classes)
    M = ACT_MISH
    # --- cspdarknet53, custom_layers.py:100-138
    x = b.conv("input", 32, 3)
    x = b.conv(x, 64, 3, down=True)
    x = b.csp(x, 64, 1, bottleneck=True)
    x = b.conv(x, 64, 1, act=M)
    x = b.conv(x, 128, 3, down=True, act=M)
    x = b.csp(x, 64, 2)
    x = b.conv(x, 128, 1, act=M)
    x = b.conv(x, 256, 3, down=True, act=M)
    x = b.csp(x, 128, 8)
    route0 = x = b.conv(x, 256, 1, act=M)
    x = b.conv(x, 512, 3, down=True, act=M)
    x = b.csp(x, 256, 8)
    route1 = x = b.conv(x, 512, 1, act=M)
    x = b.conv(x, 1024, 3, down=True, act=M)
    x = b.csp(x, 512, 4)
    x = b.conv(x, 1024, 1, act=M)
    x = b.conv(x, 512, 1)
    x = b.conv(x, 1024, 3)
    x = b.conv(x, 512, 1)
    x = b.concat(b.maxpool(x, 13), b.maxpool(x, 9), b.maxpool(x, 5), x)
    x = b.conv(x, 512, 1)
    x = b.conv(x, 1024, 3)
    route2 = b.conv(x, 512, 1)
    # --- yolov4_neck, custom_layers.py:141-198
    nout = 3 * (num_classes + 5)
    route_input = route2
    x = b.conv(route2, 256, 1)
    x = b.upsample(x)
    r1 = b.conv(route1, 256, 1)
    x = b.concat(r1, x)
    for f, k in ((256, 1), (512, 3), (256, 1), (512, 3), (256, 1)):
        x = b.conv(x, f, k)
    r1 = x
    x = b.conv(x, 128, 1)
    x = b.upsample(x)
    r0 = b.conv(route0, 128, 1)
    x = b.concat(r0, x)
    for f, k in ((128, 1), (256, 3), (128, 1), (256, 3), (128, 1)):
        x = b.conv(x, f, k)
    r0 = x
    x = b.conv(x, 256, 3)
    sb = b.conv(x, nout, 1, act=ACT_LINEAR, bn=False)
    x = b.conv(r0, 256, 3, down=True)
    x = b.concat(x, r1)
    for f, k in ((256, 1), (512, 3), (256, 1), (512, 3), (256, 1)):
        x = b.conv(x, f, k)
    r1 = x
    x = b.conv(x, 512, 3)
    mb = b.conv(x, nout, 1, act=ACT_LINEAR, bn=False)
    x = b.conv(r1, 512, 3, down=True)
    x = b.concat(x, route_input)
    for f, k in ((512, 1), (1024, 3), (512, 1), (1024, 3), (512, 1)):
        x = b.conv(x, f, k)
    x = b.conv(x, 1024, 3)
    lb = b.conv(x, nout, 1, act=ACT_LINEAR, bn=False)
    p = b.p
    p.heads = (sb, mb, lb)
    assert len(p.convs) == NUM_CONVS
    assert tuple(int(h[1:]) for h in p.heads) == HEAD_CONV_IDXS
    return p
