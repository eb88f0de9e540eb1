"""Drop-in for the inference-path names of the reference's `utils.py`
(`load_weights` :12-53, `get_detection_data` :56-78, `draw_bbox` :88-118, `voc_ap` :311-356, `read_txt_to_list` :469-475)."""
import os as _os, sys as _sys
_sys.path.insert(0, _os.path.dirname(_os.path.abspath(__file__)))
from yolo4hip.api import load_weights  # noqa: E402,F401
from yolo4hip.prepost import get_detection_data, draw_bbox  # noqa: E402,F401
from yolo4hip.evalmap import voc_ap, read_txt_to_list  # noqa: E402,F401
