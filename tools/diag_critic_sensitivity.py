"""Development aid: is a critic-gradient mismatch rounding-sized?  Per tensor: HIP vs fp64 oracle, the oracle's own fp32
vs fp64, and the fp64 oracle against itself on inputs perturbed by one part in 1e6 (conditioning of the WGAN-GP gradient:
ReLU / max-pool kinks re-route gradient paths, the penalty scales them by 1/norm)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from oracle import depgan_oracle as O
from dep_gan_im_amd import Engine

def rel(a, b):
    a, b = np.asarray(a, np.float64), np.asarray(b, np.float64)
    return float(np.abs(a - b).max() / (np.abs(b).max() + 1e-30))

img, B, seed = int(sys.argv[1]), int(sys.argv[2]), int(sys.argv[3])
PG = O.init_generator(seed, bias_std=0.05); PD1 = O.init_critic(seed + 1, bias_std=0.05, img=img); PD2 = O.init_critic(seed + 2, bias_std=0.05, img=img)
x, y2, z, ep = O.synth_batch(seed + 5, B, img, img)
rng = np.random.default_rng(seed)
x = (x + 0.02 * rng.uniform(size=x.shape)).astype(np.float32); y2 = (y2 + 0.02 * rng.uniform(size=y2.shape)).astype(np.float32)
eng = Engine(B, img, img, 1)
for n, P in (("G", PG), ("D_y2", PD1), ("D_dem", PD2)): eng.set_weights(n, P)
for which, PD, key in (("D_y2", PD1, "y2"), ("D_dem", PD2, "dem")):
    out = eng.critic(which, y2, x, z, ep, update=False)
    gg = eng.get_grads(which)
    o64, g64, a64 = O.critic_grads(PD, PG, y2, x, z, ep, key, dtype=torch.float64)
    o32, g32, _ = O.critic_grads(PD, PG, y2, x, z, ep, key, dtype=torch.float32)
    y2p = (y2.astype(np.float64) * (1 + 1e-6)).astype(np.float32)
    _, gp, _ = O.critic_grads(PD, PG, y2p, x, z, ep, key, dtype=torch.float64)
    print(which, "outs", out, o64, "norms", a64["norm"])
    for k in g64:
        print("  %-28s hip-vs-64 %.2e   f32-vs-64 %.2e   64(perturbed 1e-6)-vs-64 %.2e" % (k, rel(gg[k], g64[k]), rel(g32[k], g64[k]), rel(gp[k], g64[k])))
