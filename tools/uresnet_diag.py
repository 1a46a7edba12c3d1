"""Per-tensor comparison of the DEP-UResNet phase-1 step against the oracle (run on the GPU box)."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from oracle import depgan_oracle as O
from dep_gan_im_amd import Engine

img = int(sys.argv[1]) if len(sys.argv) > 1 else 64
B = int(sys.argv[2]) if len(sys.argv) > 2 else 4
seed = 7
drop = int(sys.argv[3]) if len(sys.argv) > 3 else 12345
P = O.init_generator(seed, nc_out=4, randomize_bn=True, bias_std=0.05)
hs = float(sys.argv[4]) if len(sys.argv) > 4 else 1.0
P["gen_segmentation/kernel"] = (P["gen_segmentation/kernel"] * hs).astype(np.float32)
x, z, lab = O.synth_uresnet_batch(seed + 3, B, img, img)
eng = Engine(B, img, img, 1, lrG=1e-4, beta1=0.9, beta2=0.999, nc_out=4)
eng.set_weights("G", P)

def rel(a, b):
    a, b = np.asarray(a, np.float64), np.asarray(b, np.float64)
    return float(np.abs(a - b).max() / (np.abs(b).max() + 1e-30))

p_ref = O.uresnet_predict(P, x, z)
p = eng.g_forward(x, z).cpu().numpy()
print("predict rel", rel(p, p_ref), "sum", p.sum(-1).min(), p.sum(-1).max())
print("eval loss", eng.uresnet(x, z, lab, "eval"),
      float(O.keras_categorical_crossentropy_t(torch.tensor(p_ref), torch.tensor(lab))))

loss32, g32, st32 = O.uresnet_grads(P, x, z, lab, drop_seed=drop or None)
loss64, g64, st64 = O.uresnet_grads(P, x, z, lab, drop_seed=drop or None, dtype=torch.float64)
loss = eng.uresnet(x, z, lab, "grads", drop_seed=drop)
print("loss gpu %.6f oracle32 %.6f oracle64 %.6f" % (loss, loss32, loss64))
G = eng.get_grads("G")
worst = []
for k in g64:
    if np.abs(g64[k]).max() < 1e-9:      # bias in front of a batch-statistics BN: exactly zero gradient
        print("zero-grad %-40s gpu max|g| %.2e oracle32 %.2e" % (k, np.abs(G[k]).max(), np.abs(g32[k]).max()))
        continue
    r, r32 = rel(G[k], g64[k]), rel(g32[k], g64[k])
    worst.append((r, r32, k))
worst.sort(reverse=True)
for r, r32, k in worst[:25]:
    print("%-44s gpu-vs-64 %.3e   oracle32-vs-64 %.3e" % (k, r, r32))
def cat(d):
    return np.concatenate([np.asarray(d[k], np.float64).reshape(-1) for k in g64])
a, b, c = cat(G), cat(g32), cat(g64)
print("global L2 rel: gpu %.3e oracle32 %.3e" % (np.linalg.norm(a - c) / np.linalg.norm(c),
                                                np.linalg.norm(b - c) / np.linalg.norm(c)))
l2 = sorted(((np.linalg.norm(np.float64(G[k]) - g64[k]) / (np.linalg.norm(g64[k]) + 1e-30),
              np.linalg.norm(np.float64(g32[k]) - g64[k]) / (np.linalg.norm(g64[k]) + 1e-30), k)
             for k in g64 if np.abs(g64[k]).max() >= 1e-9), reverse=True)
print("per-tensor L2 max: gpu %.3e (%s) oracle32 %.3e" % (l2[0][0], l2[0][2], max(t[1] for t in l2)))
print("max over all: gpu", worst[0][0], "oracle32", max(w[1] for w in worst))
# moving statistics after one phase-1 pass
W = eng.get_weights("G")
bad = 0
for name, (mean, var, n, fused) in st64.items():
    corr = n / (n - 1.0) if fused else n / (n - (1.0 + O.BN_EPS))
    mm = P[name + "/moving_mean"] * 0.99 + mean.numpy() * 0.01
    mv = P[name + "/moving_variance"] * 0.99 + var.numpy() * corr * 0.01
    r1, r2 = rel(W[name + "/moving_mean"], mm), rel(W[name + "/moving_variance"], mv)
    if max(r1, r2) > 1e-4:
        bad += 1
        print("moving stats", name, r1, r2)
print("moving stats mismatches:", bad)
