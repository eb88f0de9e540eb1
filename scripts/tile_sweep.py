"""Per-layer tile sweep inside a ONE-image step: the shipped latency schedule with every conv of one class (ksize, cin, cout) forced onto a
candidate tile id (split-K ids allowed), step time of 3 x 100 synchronised-at-the-end predicts, and the difference to the shipped schedule
per layer of the class -- the layer runs cold, in its place in the step (LABNOTES.md section 4.7: this is what showed a split costing 40-120 us
per layer before its fence was removed).
usage: tile_sweep.py <bf16|f32> <ksize> <cin> <cout> <tile id> [<tile id> ...]      e.g.  tile_sweep.py f32 3 512 1024 48 49 148 149 249"""
import json, os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "yolo-v4-tf.keras_amd")); sys.path.insert(0, ROOT)
import torch
from yolo4hip import weights as W, ext
from yolo4hip.config import make_config
from yolo4hip.engine import Engine
from yolo4hip.plan import build_plan
dtype = sys.argv[1]; ks, cin, cout = int(sys.argv[2]), int(sys.argv[3]), int(sys.argv[4]); cands = [int(x) for x in sys.argv[5:]]
size = 608
plan = build_plan(size, 80)
eng = Engine(80, make_config(size), max_batch=1, dtype=dtype, alias_workspace=True)
eng.load_weight_blob(W.flatten(W.synth_weights(plan, 0)))
eng.ensure_schedule(tune=False, verbose=False)
eng.set_splitk(True)
imgs = torch.from_numpy(W.synth_images(1, size, 0)).to(eng.device)
outs = eng.alloc_outputs(1)
tab = eng.layer_table()
idx = [i for i, L in enumerate(tab) if L["ksize"] == ks and L["cin"] == cin and L["cout"] == cout]
print("convs", idx, [(tab[i]["ksize"], tab[i]["cin"], tab[i]["cout"]) for i in idx[:1]])
base = json.load(open(os.path.join(os.path.dirname(W.__file__), "schedules", "608_80_1_%s.json" % dtype)))["tiles"]
print("shipped tiles there:", [base[i] for i in idx])
def t_of(tiles):
    eng.set_tiles(tiles)
    for _ in range(10): eng.predict_device(imgs, outs)
    torch.cuda.synchronize(); best = 1e9
    for _ in range(3):
        t0 = time.perf_counter()
        for _ in range(100): eng.predict_device(imgs, outs)
        torch.cuda.synchronize(); best = min(best, (time.perf_counter() - t0) * 10)
    return best
t0 = t_of(base)
print("shipped: %.4f ms" % t0)
for c in cands:
    tl = list(base)
    for i in idx: tl[i] = c
    try:
        t = t_of(tl)
        print("tile %4d: %.4f ms  (%+.1f us per layer)" % (c, t, (t - t0) * 1e3 / len(idx)))
    except ext.Y4Error as e:
        print("tile %4d: refused" % c)
