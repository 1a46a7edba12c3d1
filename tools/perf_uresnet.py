"""Config 5 (DEP-UResNet supervised step, batch 32, 256x256): step time and per-class split."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import dep_gan_im_amd as dg
B = int(sys.argv[1]) if len(sys.argv) > 1 else 32
dev = torch.device("cuda:0")
eng = dg.Engine(B, 256, 256, 1, lrG=1e-4, beta1=0.9, beta2=0.999, nc_out=4)
rng = np.random.default_rng(0)
x = torch.from_numpy(rng.standard_normal((B, 256, 256, 1)).astype(np.float32)).to(dev)
z = torch.from_numpy(rng.standard_normal((B, 32, 1)).astype(np.float32)).to(dev)
lab = torch.from_numpy(np.eye(4, dtype=np.float32)[rng.integers(0, 4, (B, 256, 256))]).to(dev)
for i in range(3): eng.uresnet(x, z, lab, "step", drop_seed=i + 1)
torch.cuda.synchronize(); t0 = time.perf_counter()
N = 10
for i in range(N): loss = eng.uresnet(x, z, lab, "step", drop_seed=10 + i)
torch.cuda.synchronize(); ms = (time.perf_counter() - t0) / N * 1e3
print("DEP-UResNet train_on_batch B=%d: %.2f ms/step  %.1f slices/s  %.1f TF/s (70.6 GFLOP/slice)  loss %.4f" % (B, ms, B / ms * 1e3, 70.6e9 * B / ms / 1e9, loss))
eng.profile(True); eng.profile_reset()
for i in range(3): eng.uresnet(x, z, lab, "step", drop_seed=30 + i)
for k, nm in ((0, "mfma conv"), (1, "mfma wgrad"), (2, "other")):
    t, n, fl = eng.profile_read(k)
    print("  class %-10s %7.2f ms/step  %4d launches  %6.1f TF/s" % (nm, t / 3, n // 3, fl / t / 1e9 if t else 0))
if len(sys.argv) > 2:   # per-label table of the `other` class: python tools/perf_uresnet.py 32 <csv>
    import csv, collections
    eng.profile_dump(sys.argv[2])
    acc = collections.defaultdict(lambda: [0, 0.0])
    for r in csv.DictReader(open(sys.argv[2])):
        if int(r["class"]) == 2:
            a = acc[r["label"]]; a[0] += 1; a[1] += float(r["ms"])
    for k, v in sorted(acc.items(), key=lambda kv: -kv[1][1])[:25]:
        print("  other %-34s %3d launches/step %7.3f ms/step" % (k, v[0] // 3, v[1] / 3))
