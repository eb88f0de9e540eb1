import sys, json, os, time
sys.path.insert(0,'yolo-v4-tf.keras_amd')
import torch
from yolo4hip import weights as W
from yolo4hip.config import make_config
from yolo4hip.engine import Engine
from yolo4hip.plan import build_plan
for dtype in ('bf16','f32'):
    flat = W.flatten(W.synth_weights(build_plan(608, 80), 0))
    for name in sys.argv[1:]:
        eng = Engine(80, make_config(608), max_batch=1, dtype=dtype, alias_workspace=True)
        eng.load_weight_blob(flat)
        saved=json.load(open(name.replace('DT',dtype)))
        if dtype!='f32':
            eng.set_stem_fusion(True); eng.set_chain_fusion(True); eng.set_stage_fusion(True); eng.set_res_fusion(True)
        eng.apply_schedule(saved)
        imgs = torch.from_numpy(W.synth_images(1, 608, 0)).to(eng.device)
        outs = eng.alloc_outputs(1)
        for _ in range(20): eng.predict_device(imgs, outs)
        torch.cuda.synchronize()
        best=1e9
        for r in range(5):
            t=time.perf_counter()
            for _ in range(200): eng.predict_device(imgs, outs)
            torch.cuda.synchronize()
            best=min(best,(time.perf_counter()-t)/200*1e3)
        print(dtype, name, f"{best:.4f} ms")
        eng.close()
