"""Inference constants of the path (mirror of the reference's module-global dict, `config.py:1-17`).

Only the keys the inference path reads are kept; the training knobs (`iou_loss_thresh`,
`batch_size`) are carried so that code doing `yolo_config['batch_size']` keeps working, but nothing
here consumes them.  Unlike the reference (`models.py:26-37` ignores the ctor's `config=` argument),
`Yolov4(config=...)` honours a passed dict.
"""

yolo_config = {
    # Basic
    'img_size': (416, 416, 3),
    'anchors': [12, 16, 19, 36, 40, 28, 36, 75, 76, 55, 72, 146, 142, 110, 192, 243, 459, 401],
    'strides': [8, 16, 32],
    'xyscale': [1.2, 1.1, 1.05],

    # Training (unused by this inference-only framework)
    'iou_loss_thresh': 0.5,
    'batch_size': 8,
    'num_gpu': 1,

    # Inference
    'max_boxes': 100,
    'iou_threshold': 0.413,
    'score_threshold': 0.3,
}


def make_config(img_size=416, **overrides):
    """Copy of `yolo_config` at another square resolution (e.g. 608 for the headline configs)."""
    cfg = dict(yolo_config)
    cfg['img_size'] = (int(img_size), int(img_size), 3)
    cfg.update(overrides)
    return cfg
