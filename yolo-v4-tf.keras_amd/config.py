"""Drop-in for the reference's `config.py` (`from config import yolo_config`)."""
import os as _os, sys as _sys
_sys.path.insert(0, _os.path.dirname(_os.path.abspath(__file__)))
from yolo4hip.config import yolo_config, make_config  # noqa: E402,F401
