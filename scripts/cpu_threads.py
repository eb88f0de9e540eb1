"""Which torch thread count is fastest for the CPU oracle on this host (for bench.py's cpu_baseline)."""
import sys, os, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "yolo-v4-tf.keras_amd")); sys.path.insert(0, ROOT)
import torch
from yolo4hip import weights as W
from yolo4hip.plan import build_plan
from oracle import forward as OF
plan = build_plan(608, 80); ws = W.synth_weights(plan, 0); imgs = W.synth_images(2, 608, 0)
print("cpu_count", os.cpu_count(), "affinity", len(os.sched_getaffinity(0)))
for t in [int(a) for a in sys.argv[1:]]:
    torch.set_num_threads(t)
    OF.yolo_model_forward(imgs[:1], ws, 80)
    t0 = time.perf_counter(); OF.yolo_model_forward(imgs, ws, 80); dt = time.perf_counter() - t0
    print(f"threads {t}: {2/dt:.3f} img/s ({dt:.2f} s for 2 images)", flush=True)
