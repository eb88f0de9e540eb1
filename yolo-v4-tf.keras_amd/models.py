"""Drop-in for the reference's `models.py`: `from models import Yolov4` (reference README.md:26-31)."""
import os as _os, sys as _sys
_sys.path.insert(0, _os.path.dirname(_os.path.abspath(__file__)))
from yolo4hip.api import Yolov4, load_weights  # noqa: E402,F401
from yolo4hip.config import yolo_config  # noqa: E402,F401
