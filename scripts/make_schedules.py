"""Tunes the schedules that ship with the package (yolo4hip/schedules/<side>_<classes>_<batch>_<dtype>.json) on the GPU box:

    python scripts/make_schedules.py gpurun_out/schedules [608_80_1_bf16 416_80_32_bf16 ...]

One `Engine.ensure_schedule` tuning run per shape (all fusions on, split-K offered for latency-sized engines), written to the
output directory under the shipped name; copy the files into yolo-v4-tf.keras_amd/yolo4hip/schedules/ to ship them.  The
headline shape's file (608_80_32_bf16) is NOT made here: it is the voted two-in-flight schedule of scripts/vote_schedule.py."""
import json
import os
import shutil
import sys
import tempfile
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "yolo-v4-tf.keras_amd")); sys.path.insert(0, ROOT)
import torch
from yolo4hip import weights as W
from yolo4hip.config import make_config
from yolo4hip.engine import Engine
from yolo4hip.plan import build_plan

out = sys.argv[1]
shapes = sys.argv[2:] or ["416_80_32_bf16", "416_80_1_bf16", "608_80_1_bf16", "608_80_1_f32", "416_80_1_f32", "416_80_32_f32",
                         "608_80_32_f32"]      # ..._32_f32: the facade's own defaults (Yolov4(): f32, max_batch 32)
os.makedirs(out, exist_ok=True)
os.environ["YOLO4HIP_CACHE"] = tempfile.mkdtemp(prefix="y4sched_")
os.environ.setdefault("YOLO4HIP_LATENCY", "1")      # the batch-1 schedules that ship carry split-K ids (tested against the oracle)
flats = {}
for key in shapes:
    side, ncls, batch, dtype = key.split("_")
    side, ncls, batch = int(side), int(ncls), int(batch)
    if ncls not in flats:
        flats[ncls] = W.flatten(W.synth_weights(build_plan(608, ncls), 0))      # fully convolutional: one weight set per class count
    eng = Engine(ncls, make_config(side), max_batch=batch, dtype=dtype, alias_workspace=True)
    eng.load_weight_blob(flats[ncls])
    if os.path.exists(os.path.join(os.path.dirname(W.__file__), "schedules", key + ".json")) and os.environ.get("Y4_RETUNE") != "1":
        print(key, "ships already (Y4_RETUNE=1 to tune it again)")
        eng.close()
        continue
    eng.shipped_schedule = lambda: None            # (re)tune even when a file ships for this shape
    t0 = time.perf_counter()
    src, path = eng.ensure_schedule(tune=True, verbose=False)
    assert src == "tuned", (key, src)
    saved = json.load(open(path))
    json.dump(saved, open(os.path.join(out, key + ".json"), "w"))
    imgs = torch.from_numpy(W.synth_images(batch, side, 0)).to(eng.device)
    outs = eng.alloc_outputs(batch)
    for _ in range(5): eng.predict_device(imgs, outs)
    torch.cuda.synchronize()
    t1 = time.perf_counter()
    n = 50
    for _ in range(n): eng.predict_device(imgs, outs)
    torch.cuda.synchronize()
    ms = (time.perf_counter() - t1) / n * 1e3
    print(f"{key}: tuned in {t1 - t0:.1f} s, {ms:.3f} ms per predict of {batch} image(s), split-K ids: "
          f"{sorted(set(t for t in saved['tiles'] if t >= 100))}, stage {saved['stage_fusion']}, res mask {saved['res_fusion_mask']}", flush=True)
    eng.close()
shutil.rmtree(os.environ["YOLO4HIP_CACHE"], ignore_errors=True)
