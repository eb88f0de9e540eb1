"""Debug aid: compare conv outputs with and without chain fusion (GPU)."""
import sys, os
sys.path.insert(0, os.path.join(os.path.dirname(__file__), "..", "yolo-v4-tf.keras_amd"))
sys.path.insert(0, os.path.join(os.path.dirname(__file__), ".."))
import numpy as np
from tests.test_gpu_forward import _setup

size, n, dtype = int(sys.argv[1]), int(sys.argv[2]), sys.argv[3]
cfg, plan, ws, imgs, eng = _setup(size, 3, n, dtype, seed=6)
heads = eng.forward_heads(imgs)
idxs = (2, 3, 4, 5, 6, 7, 8, 9, 10, 11, 12, 13, 14, 15, 16, 17)
ref = {i: eng.conv_output(i, n) for i in idxs}
print("chains", eng.set_chain_fusion(True))
fh = eng.forward_heads(imgs)
for i in idxs:
    got = eng.conv_output(i, n)
    d = np.abs(got - ref[i])
    print(i, got.shape, "max", d.max(), "mean", d.mean(), "exact frac", (d == 0).mean(), "ref absmax", np.abs(ref[i]).max())
for a, b in zip(heads, fh):
    print("head diff max", np.abs(a - b).max(), "q999", np.quantile(np.abs(a - b), 0.999))
