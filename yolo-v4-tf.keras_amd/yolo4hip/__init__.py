"""yolo4hip -- host-side mirror of the taipingeric/yolo-v4-tf.keras inference API over libyolo4hip.so
(hand-written gfx950 HIP kernels).  Importing the package needs neither a GPU nor the shared library;
`Engine` / `Yolov4` raise if either is missing -- there is no CPU fallback."""
from .config import yolo_config, make_config  # noqa: F401
from .plan import build_plan  # noqa: F401

__all__ = ["yolo_config", "make_config", "build_plan", "Yolov4", "Engine"]


def __getattr__(name):
    if name == "Yolov4":
        from .api import Yolov4
        return Yolov4
    if name == "Engine":
        from .engine import Engine
        return Engine
    raise AttributeError(name)
