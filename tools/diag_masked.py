"""Diagnostic: per-tensor errors of the generator gradient under HIP's masks, HIP vs fp64 and the oracle's own fp32
run vs fp64 (same masks), worst tensors first.  usage: python tools/diag_masked.py [img] [seed]"""
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import test_gpu_masked as T   # noqa: E402
from oracle import manual as M   # noqa: E402

img = int(sys.argv[1]) if len(sys.argv) > 1 else 256
seed = int(sys.argv[2]) if len(sys.argv) > 2 else 231
B = 2
PG, PD1, PD2, x, y2, z, ep = T.setup(img, B, seed, noisy=True, trained_regime=True)
eng = T.engine(img, B, PG, PD1, PD2)
eng.generator(x, y2, z, "grads")
gg = eng.get_grads("G")
masks = T.hip_generator_masks(eng, x, y2, B)
_, g64 = M.g_grads_manual(PG, PD1, PD2, x, y2, z, masks=masks)
_, g32 = M.g_grads_manual(PG, PD1, PD2, x, y2, z, masks=masks, dtype=torch.float32)
e_hip, e_32 = T.tensor_errors(gg, g64), T.tensor_errors(g32, g64)
for k in sorted(e_hip, key=e_hip.get, reverse=True)[:12]:
    print("%-40s HIP %.2e  oracle-fp32 %.2e  max|g| %.3e" % (k, e_hip[k], e_32[k], np.abs(g64[k]).max()))
k = max(e_hip, key=e_hip.get)
d = np.abs(gg[k].astype(np.float64) - g64[k]).reshape(-1)
print(k, "entries", d.size, "worst idx", int(d.argmax()), "got", gg[k].reshape(-1)[d.argmax()], "want", g64[k].reshape(-1)[d.argmax()])
print("sorted abs errors (top 8):", np.sort(d)[::-1][:8])
