"""A/B of two sets of one-image schedules on ONE box, alternating: the shipped yolo4hip/schedules/<shape>.json against the files of the same
name in <dir> (e.g. the output of scripts/make_schedules.py), 3 rounds x 100 predicts each, ms per predict.
usage: sched_ab.py <dir with 608_80_1_bf16.json 608_80_1_f32.json 416_80_1_bf16.json 416_80_1_f32.json>"""
import json, os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "yolo-v4-tf.keras_amd")); sys.path.insert(0, ROOT)
import torch
from yolo4hip import weights as W
from yolo4hip.config import make_config
from yolo4hip.engine import Engine
from yolo4hip.plan import build_plan
newdir = sys.argv[1]
flat = W.flatten(W.synth_weights(build_plan(608, 80), 0))
for key in ["608_80_1_bf16", "608_80_1_f32", "416_80_1_bf16", "416_80_1_f32"]:
    side, ncls, batch, dtype = key.split("_"); side, ncls, batch = int(side), int(ncls), int(batch)
    eng = Engine(ncls, make_config(side), max_batch=batch, dtype=dtype, alias_workspace=True)
    eng.load_weight_blob(flat)
    imgs = torch.from_numpy(W.synth_images(batch, side, 0)).to(eng.device)
    outs = eng.alloc_outputs(batch)
    res = {}
    for rnd in range(3):
        for name, path in (("shipped", os.path.join(os.path.dirname(W.__file__), "schedules", key + ".json")), ("new", os.path.join(newdir, key + ".json"))):
            if dtype != "f32":
                eng.set_stem_fusion(True); eng.set_chain_fusion(True); eng.set_stage_fusion(True); eng.set_res_fusion(True)
            eng.apply_schedule(json.load(open(path)))
            for _ in range(10): eng.predict_device(imgs, outs)
            torch.cuda.synchronize(); t0 = time.perf_counter()
            for _ in range(100): eng.predict_device(imgs, outs)
            torch.cuda.synchronize(); res.setdefault(name, []).append((time.perf_counter() - t0) * 10)
    print(key, {k: [round(x, 4) for x in v] for k, v in res.items()})
    eng.close()
