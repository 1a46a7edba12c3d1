"""Per-launch timing of the generator forward alone (north_star's >= 70 % target), sorted by time."""
import os, sys, csv, collections
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import dep_gan_im_amd as dg
from bench import synth

B = int(sys.argv[1]) if len(sys.argv) > 1 else 32
dev = torch.device("cuda:0")
eng = dg.Engine(B, 256, 256, 1)
x, y2, z, ep = [torch.from_numpy(a).to(dev) for a in synth(1000, B)]
for _ in range(3): eng.g_forward(x, z)
torch.cuda.synchronize()
import time
t0 = time.perf_counter()
for _ in range(10): eng.g_forward(x, z)
torch.cuda.synchronize()
ms = (time.perf_counter() - t0) / 10 * 1e3
print("G forward %.3f ms  %.1f TF/s  %.1f %% of 157.3" % (ms, 23.513e9 * B / ms / 1e9, 23.513e9 * B / ms / 1e9 / 1.573))
eng.profile(True); eng.profile_reset()
N = 5
for _ in range(N): eng.g_forward(x, z)
path = "gpurun_out/gfwd.csv"
eng.profile_dump(path)
agg = collections.OrderedDict()
for r in csv.DictReader(open(path)):
    k = (r["class"], r["label"])
    a = agg.setdefault(k, [0, 0.0, 0.0])
    a[0] += 1; a[1] += float(r["ms"]); a[2] += float(r["gflop"])
tot = sum(a[1] for a in agg.values())
print("sum of kernels %.3f ms" % (tot / N))
for (k, l), a in sorted(agg.items(), key=lambda kv: -kv[1][1]):
    print("c%s %-34s n %5.1f  %7.3f ms  %6.1f TF/s  %4.1f%%" % (k, l, a[0] / N, a[1] / N, a[2] / a[1] if a[1] else 0, 100 * a[1] / tot))
