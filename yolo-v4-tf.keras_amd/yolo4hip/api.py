"""`Yolov4` facade: the reference's user API (models.py:17-127, 509-529) over the HIP engine.

Drop-in surface kept: constructor signature and asserts, the attributes users read
(`class_names, img_size, num_classes, anchors, xyscale, strides, output_sizes, class_color, config,
weight_path, max_boxes`), `.yolo_model.predict(imgs)` -> 3 raw heads, `.inference_model.predict(imgs)` ->
`[boxes, scores, classes, valid_detections]`, `.predict / .predict_img / .predict_raw / .predict_nonms`
with their prints (`img shape:`, `# of bboxes:`) and DataFrame layout.
Differences, all supersets: the passed `config` is honoured (the reference reads the module global,
models.py:26-37); grid sizes follow img_size (the reference hard-codes 52/26/13); extra keyword-only
arguments pick the compute dtype / batch capacity.  `save_model` / `load_model` (models.py:86-93) are stand-ins on a
self-describing checkpoint of this framework (Keras' SavedModel / H5 needs TensorFlow); `eval_map` (models.py:182-507)
is the VOC mAP tool over the exported text files (yolo4hip/evalmap.py).  Training (`fit`, `training_model`) is out of
scope for this inference framework and raises.
"""
import json
import os

import numpy as np

from . import prepost, weights as W
from .config import yolo_config
from .engine import Engine
from .plan import build_plan


class _KerasLikeModel:
    """The slice of `tf.keras.Model` the reference touches: `.predict(x)`."""

    def __init__(self, fn, name):
        self._fn, self.name = fn, name

    def predict(self, x, batch_size=None, verbose=0, **_):
        return self._fn(x)

    __call__ = predict


class Yolov4(object):
    def __init__(self, weight_path=None, class_name_path='coco_classes.txt', config=yolo_config, *,
                 dtype='f32', max_batch=32, synth_seed=0, device=None, device_preprocess=True, tune=None, share_schedule=None):
        assert config['img_size'][0] == config['img_size'][1], 'not support yet'
        assert config['img_size'][0] % config['strides'][-1] == 0, 'must be a multiple of last stride'
        self.class_names = [line.strip() for line in open(class_name_path).readlines()]
        self.img_size = tuple(config['img_size'])
        self.num_classes = len(self.class_names)
        self.weight_path = weight_path
        self.anchors = np.array(config['anchors']).reshape((3, 3, 2))
        self.xyscale = config['xyscale']
        self.strides = config['strides']
        self.output_sizes = [self.img_size[0] // s for s in self.strides]
        self.class_color = {name: list(np.random.random(size=3) * 255) for name in self.class_names}
        self.max_boxes = config['max_boxes']
        self.iou_loss_thresh = config.get('iou_loss_thresh', 0.5)
        self.config = config
        assert self.num_classes > 0, 'no classes detected!'
        self._dtype, self._max_batch, self._synth_seed, self._device = dtype, max_batch, synth_seed, device
        self._device_preprocess = device_preprocess
        # A shape without a shipped / cached schedule is tuned ONCE (a full-size predict + y4_autotune on `max_batch` synthetic
        # images: seconds at 160^2, ~20 s at 608^2 x 32; written to $YOLO4HIP_CACHE, default ~/.cache/yolo4hip):
        #   tune=None (default)  on the first predict of this object, not in the constructor; YOLO4HIP_TUNE=0: never
        #   tune=True            now, in the constructor          tune=False   never (built-in tile heuristic)
        # share_schedule=True (or YOLO4HIP_SHARE_SCHEDULE=1) under an initialised torch.distributed group of more than one rank makes
        # the constructor a COLLECTIVE -- rank 0 resolves (and, if allowed, tunes) the schedule now and broadcasts it, so every rank
        # runs the same tile set after ONE tuning run; every rank must then construct the same model.  Off by default (a
        # model built on some ranks only -- a rank-0 evaluation inside a DDP job -- must not block in a broadcast): each rank then
        # resolves its own schedule, which for shipped and cached schedules is the same file everywhere.
        self._share_schedule = (os.environ.get('YOLO4HIP_SHARE_SCHEDULE', '0') == '1') if share_schedule is None else bool(share_schedule)
        self._tune = (os.environ.get('YOLO4HIP_TUNE', '1') != '0') if tune is None else bool(tune)
        self._tune_eager = tune is True
        self.build_model(load_pretrained=True if self.weight_path else False)

    def build_model(self, load_pretrained=True):
        self.plan = build_plan(self.img_size[0], self.num_classes)
        # alias_workspace: like the reference's Keras model, the facade keeps no intermediate activations (2.7x less HBM)
        self.engine = Engine(self.num_classes, self.config, max_batch=self._max_batch, dtype=self._dtype,
                             device=self._device, alias_workspace=True)
        self.yolo_model = _KerasLikeModel(self._tuned_first(self.engine.forward_heads), 'yolo_model')
        print(f"nms iou: {self.config['iou_threshold']} score: {self.config['score_threshold']}")
        self.inference_model = _KerasLikeModel(self._tuned_first(self.engine.predict), 'inference_model')
        self.training_model = None      # training is out of scope (inference-only framework)
        if load_pretrained and self.weight_path and self.weight_path.endswith('.weights'):
            load_weights(self, self.weight_path)
            print(f'load from {self.weight_path}')
        else:
            # the reference leaves Keras' random initialisation in place; ours is the seeded synthetic set
            self._set_weights(W.flatten(W.synth_weights(self.plan, self._synth_seed)))
        # No silent un-tuned shapes: the tuned schedule that ships for this (size, classes, batch, dtype), else the one
        # tuned on this machine before, else tune -- once, cached on disk -- and say which is active.  Schedules without split-K
        # ids are bit-identical scheduling choices (tests/test_gpu_forward.py), so WHEN the tuning happens changes no result.
        from . import dist as D
        self._tune_pending = False
        if self._share_schedule and D.group_rank_world()[1] > 1:
            self.schedule_source = self.engine.ensure_schedule(tune=self._tune, share=True)
        elif self._tune_eager or not self._tune:
            self.schedule_source = self.engine.ensure_schedule(tune=self._tune)
        else:
            self.schedule_source = self.engine.ensure_schedule(tune=False, verbose=False)
            self._tune_pending = self.schedule_source[0] == 'heuristic'
            if self._tune_pending:
                print('schedule: none ships or is cached for this shape -- tuned on the first predict (tune=True: in the constructor)')
            else:
                self.engine.say_schedule()

    def _ensure_tuned(self):
        if self._tune_pending:
            self._tune_pending = False
            self.schedule_source = self.engine.ensure_schedule(tune=True)

    def _tuned_first(self, fn):
        def call(x, **kw):
            self._ensure_tuned()
            return fn(x, **kw)
        return call

    def _set_weights(self, flat):
        """flat: the Darknet-order float32 stream (weights.flatten).  Kept on the host like Keras keeps its variables,
        so that `save_model` can write it back."""
        self._flat = np.ascontiguousarray(flat, dtype=np.float32)
        self.engine.load_weight_blob(self._flat)

    # ---- reference models.py:86-93.  Keras writes architecture + variables of `yolo_model` (SavedModel / H5); without
    # TensorFlow the stand-in is a self-describing checkpoint of THIS framework: magic, a JSON header (what fixes the
    # architecture here: img_size, num_classes, anchors, strides, xyscale), then the float32 weight stream in Darknet
    # order -- dtype independent, so a model saved from a bf16 engine loads into an fp32 one.  A path ending in
    # `.weights` is written / read as a plain Darknet file instead.
    _CKPT_MAGIC = b'Y4CKPT1\n'

    def save_model(self, path):
        if path.endswith('.weights'):
            W.write_darknet(path, W.unflatten(self.plan, self._flat))
            return
        head = json.dumps({'format': 'yolo4hip-checkpoint', 'version': 1, 'img_size': list(self.img_size),
                           'num_classes': self.num_classes, 'class_names': self.class_names,
                           'anchors': np.asarray(self.anchors).reshape(-1).tolist(), 'strides': list(self.strides),
                           'xyscale': list(self.xyscale), 'n_floats': int(self._flat.size)}).encode()
        with open(path, 'wb') as fh:
            fh.write(self._CKPT_MAGIC)
            fh.write(len(head).to_bytes(8, 'little'))
            fh.write(head)
            self._flat.tofile(fh)

    def load_model(self, path):
        """Like the reference, rebuilds `yolo_model` and `inference_model` from the file; like it (models.py:90-91: `nms`
        called without thresholds) the inference model then uses the DEFAULT thresholds iou 0.413 / score 0.3."""
        if path.endswith('.weights'):
            ws, _header, _unread = W.read_darknet(path, self.plan)
            flat = W.flatten(ws)
        else:
            with open(path, 'rb') as fh:
                if fh.read(len(self._CKPT_MAGIC)) != self._CKPT_MAGIC:
                    raise ValueError(f'{path}: not a yolo4hip checkpoint (Keras SavedModel / H5 files need TensorFlow)')
                head = json.loads(fh.read(int.from_bytes(fh.read(8), 'little')).decode())
                flat = np.fromfile(fh, dtype=np.float32)
            same = (head['num_classes'] == self.num_classes and list(head['img_size']) == list(self.img_size)
                    and head['n_floats'] == self.plan.n_params)
            if not same or flat.size != self.plan.n_params:
                raise ValueError(f"{path}: checkpoint is for img_size {head['img_size']}, {head['num_classes']} classes, "
                                 f"{flat.size} floats; this model has {list(self.img_size)}, {self.num_classes}, "
                                 f"{self.plan.n_params}")
        self._set_weights(flat)
        self.yolo_model = _KerasLikeModel(self._tuned_first(self.engine.forward_heads), 'yolo_model')
        self.inference_model = _KerasLikeModel(
            self._tuned_first(lambda x: self.engine.predict(x, iou_threshold=0.413, score_threshold=0.3)), 'inference_model')

    # ---- out of scope (training)
    def fit(self, *a, **k):
        raise NotImplementedError('training is out of scope of the MI355X inference path')

    # ---- reference models.py:95-98
    def preprocess_img(self, img):
        return prepost.preprocess_img(img, self.img_size)

    # ---- reference models.py:109-123 (raw_img: RGB)
    def predict_img(self, raw_img, random_color=True, plot_img=True, figsize=(10, 10), show_text=True,
                    return_output=False):
        print('img shape: ', raw_img.shape)
        if self._device_preprocess and getattr(raw_img, 'dtype', None) == np.uint8:
            # same arithmetic as preprocess_img (tests: bit-identical), on the GPU: the uint8 image crosses PCIe
            imgs = self.engine.preprocess_u8(np.ascontiguousarray(raw_img))
        else:
            img = self.preprocess_img(raw_img)
            imgs = np.expand_dims(img, axis=0)
        pred_output = self.inference_model.predict(imgs)
        detections = prepost.get_detection_data(img=raw_img, model_outputs=pred_output, class_names=self.class_names)
        output_img = prepost.draw_bbox(raw_img, detections, cmap=self.class_color, random_color=random_color,
                                       figsize=figsize, show_text=show_text, show_img=plot_img)
        if return_output:
            return output_img, detections
        return detections

    # ---- reference models.py:125-127
    def predict(self, img_path, random_color=True, plot_img=True, figsize=(10, 10), show_text=True):
        raw_img = prepost.imread_rgb(img_path)
        return self.predict_img(raw_img, random_color, plot_img, figsize, show_text)

    # ---- reference models.py:129-139: annotation lines "path x1,y1,x2,y2,cls ..." -> one "<name> x1 y1 x2 y2" file each
    def export_gt(self, annotation_path, gt_folder_path):
        with open(annotation_path) as fh:
            for line in fh:
                fields = line.split(' ')
                stem = fields[0].split(os.sep)[-1].split('.')[0]
                with open(os.path.join(gt_folder_path, stem + '.txt'), 'w') as out:
                    for obj in fields[1:]:
                        x_min, y_min, x_max, y_max, class_id = [float(v) for v in obj.strip().split(',')]
                        out.write(f'{self.class_names[int(class_id)]} {x_min} {y_min} {x_max} {y_max}\n')

    # ---- reference models.py:141-179: the reference's only BATCHED caller of inference_model.predict.
    # One "<class> <score> <x1> <y1> <x2> <y2>" line per detection, coordinates in raw-image pixels.  Like the
    # reference it feeds cv2.imread's BGR channel order (no flip, unlike predict()).
    def export_prediction(self, annotation_path, pred_folder_path, img_folder_path, bs=2):
        with open(annotation_path) as fh:
            img_paths = [os.path.join(img_folder_path, line.split(' ')[0].split(os.sep)[-1].strip()) for line in fh]
        try:
            from tqdm import tqdm
        except ImportError:                      # progress bar only
            def tqdm(it):
                return it
        for start in tqdm(range(0, len(img_paths), bs)):
            paths = img_paths[start:start + bs]
            raws = [prepost.imread_rgb(pth)[:, :, ::-1] for pth in paths]
            if self._device_preprocess:
                imgs = self.engine.preprocess_u8([np.ascontiguousarray(r) for r in raws])
            else:
                imgs = np.stack([self.preprocess_img(r) for r in raws])
            b_boxes, b_scores, b_classes, b_valid = self.inference_model.predict(imgs)
            for k, pth in enumerate(paths):
                nb = int(b_valid[k])
                h, w = raws[k].shape[:2]
                boxes = b_boxes[k, :nb]
                boxes[:, [0, 2]] *= w                 # the returned arrays are writable host copies
                boxes[:, [1, 3]] *= h
                names = [self.class_names[int(c)] for c in b_classes[k, :nb]]
                stem = pth.split(os.sep)[-1].split('.')[0]
                with open(os.path.join(pred_folder_path, stem + '.txt'), 'w') as out:
                    for j in range(nb):
                        b = boxes[j]
                        out.write(f'{names[j]} {b_scores[k, j]} {b[0]} {b[1]} {b[2]} {b[3]}\n')

    # ---- reference models.py:182-507: VOC mAP over the folders export_gt / export_prediction wrote (text outputs and
    # prints kept, matplotlib windows not; also returns the numbers) -- see yolo4hip/evalmap.py
    def eval_map(self, gt_folder_path, pred_folder_path, temp_json_folder_path, output_files_path):
        from .evalmap import eval_map
        return eval_map(gt_folder_path, pred_folder_path, temp_json_folder_path, output_files_path)

    # ---- reference models.py:509-514 (note: no BGR->RGB flip there; cv2.imread order is BGR)
    def predict_raw(self, img_path):
        raw_img = prepost.imread_rgb(img_path)[:, :, ::-1]
        print('img shape: ', raw_img.shape)
        img = self.preprocess_img(raw_img)
        imgs = np.expand_dims(img, axis=0)
        return self.yolo_model.predict(imgs)

    # ---- reference models.py:516-529 (thresholds as run-time arguments)
    def predict_nonms(self, img_path, iou_threshold=0.413, score_threshold=0.1):
        raw_img = prepost.imread_rgb(img_path)[:, :, ::-1]
        print('img shape: ', raw_img.shape)
        img = self.preprocess_img(raw_img)
        imgs = np.expand_dims(img, axis=0)
        print(f'nms iou: {iou_threshold} score: {score_threshold}')
        self._ensure_tuned()
        pred_output = self.engine.predict(imgs, iou_threshold=iou_threshold, score_threshold=score_threshold)
        detections = prepost.get_detection_data(img=raw_img, model_outputs=pred_output, class_names=self.class_names)
        prepost.draw_bbox(raw_img, detections, cmap=self.class_color, random_color=True)
        return detections


def load_weights(model, weights_file_path):
    """reference utils.py:12-53: read a Darknet `.weights` file into the model.  `model` is a `Yolov4`
    (the reference passes the Keras `yolo_model`; the facade object owns the engine here)."""
    ws, _header, unread = W.read_darknet(weights_file_path, model.plan)
    model._set_weights(W.flatten(ws))
    if unread == 0:
        print('all weights read')
    else:
        print(f'failed to read  all weights, # of unread weights: {unread}')


def default_class_path(name='coco_classes.txt'):
    return os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), 'class_names', name)
