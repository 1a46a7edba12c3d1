"""Development aid: device time of the data step for one subject (256 x 256 x Z volumes resident in HBM)."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from dep_gan_im_amd import data as dg
Z = int(sys.argv[1]) if len(sys.argv) > 1 else 48
dev = torch.device("cuda:0")
n = 256 * 256 * Z
vols = [torch.rand(n, device=dev) for _ in range(7)]
for nicg in (2, 1):
    v = list(vols)
    if nicg == 1: v[1] = None
    for _ in range(3): dg.prep_subject_flat(v, (256, 256, Z), nicg, dev)
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(20): dg.prep_subject_flat(v, (256, 256, Z), nicg, dev)
    torch.cuda.synchronize(); us = (time.perf_counter() - t0) / 20 * 1e6
    rd = (7 if nicg == 2 else 6) * 4 * n; wr = (nicg + 1) * 4 * n + (8 * n if nicg == 2 else 0) * 1  # + normalise pass r/w of x
    print("nicg %d  Z %d: %.1f us per subject, %.0f slices/s, %.2f TB/s of %d MB algorithmic traffic" % (nicg, Z, us, Z / us * 1e6, (rd + wr) / us / 1e6, (rd + wr) >> 20))
