"""Development aid: bf16-pipe engine vs bf16-weights-only engine vs the oracles (forward of G and of a critic)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from oracle import depgan_oracle as O
from dep_gan_im_amd import Engine
def rel(a, b):
    a, b = np.asarray(a, np.float64), np.asarray(b, np.float64)
    return float(np.abs(a - b).max() / (np.abs(b).max() + 1e-30))
img, B, nicg = int(sys.argv[1]), 2, int(sys.argv[2])
PG = O.init_generator(57, nicg=nicg, bias_std=0.05); PD = O.init_critic(58, bias_std=0.05, img=img)
x, y2, z, ep = O.synth_batch(62, B, img, img, nicg=nicg)
outs = {}
for name, kw in (("w", dict(bf16_weights=True)), ("wa", dict(bf16_weights=True, bf16_mfma=True))):
    e = Engine(B, img, img, nicg, **kw)
    e.set_weights("G", PG); e.set_weights("D_y2", PD)
    outs[name] = (e.g_forward(x, z).cpu().numpy(), e.d_forward("D_y2", y2).cpu().numpy())
    e.close()
ow = (O.g_predict(O.round_kernels_bf16(PG), x, z, nicg=nicg), O.d_predict(O.round_kernels_bf16(PD), y2))
with O.bf16_activations():
    oq = (O.g_predict(O.round_kernels_bf16(PG), x, z, nicg=nicg), O.d_predict(O.round_kernels_bf16(PD), y2))
for i, n in enumerate(("G", "D")):
    print(n, "hip(w) vs oracle(w) %.2e | hip(wa) vs oracle(wa) %.2e | hip(wa) vs hip(w) %.2e | oracle(wa) vs oracle(w) %.2e"
          % (rel(outs["w"][i], ow[i]), rel(outs["wa"][i], oq[i]), rel(outs["wa"][i], outs["w"][i]), rel(oq[i], ow[i])))
d = np.abs(outs["wa"][0] - oq[0])
print("G: mean abs diff %.2e, 99.9th pct %.2e, max %.2e; max |ref| %.2e" % (d.mean(), np.quantile(d, 0.999), d.max(), np.abs(oq[0]).max()))
