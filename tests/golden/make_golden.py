"""Generates tests/golden/*.npz from the CPU oracle (oracle/depgan_oracle.py).

The reference itself cannot run here (Python-2 Keras/TF-1 scripts, no Keras, no
weights, no fixtures), so these vectors pin the ORACLE, not the reference:
"parity unpinned" (see oracle/__init__.py).  Run from the repo root:
    python tests/golden/make_golden.py [--full]
Inputs and weights are regenerated from seeds by the oracle's own seeded
constructors; weight checksums are stored to detect drift of those.
"""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from oracle import depgan_oracle as O  # noqa: E402

HERE = os.path.dirname(os.path.abspath(__file__))


def checks(P):
    return np.array([float(np.sum(np.asarray(v, np.float64))) for v in P.values()])


def case(name, img, B, seed):
    PG = O.init_generator(seed, bias_std=0.05)
    PD1 = O.init_critic(seed + 1, bias_std=0.05, img=img)
    PD2 = O.init_critic(seed + 2, bias_std=0.05, img=img)
    x, y2, z, ep = O.synth_batch(seed + 5, B, img, img)
    out = dict(img=img, B=B, seed=seed, wsumG=checks(PG), wsumD1=checks(PD1), wsumD2=checks(PD2),
               xsum=float(x.astype(np.float64).sum()), y2sum=float(y2.astype(np.float64).sum()))
    attr = O.g_predict(PG, x, z)
    out["attr_sum"] = float(attr.astype(np.float64).sum())
    out["attr_abs_sum"] = float(np.abs(attr.astype(np.float64)).sum())
    rng = np.random.default_rng(123)
    idx = rng.integers(0, attr.size, 256)
    out["attr_idx"], out["attr_samples"] = idx, attr.reshape(-1)[idx]
    out["d_y2"] = O.d_predict(PD1, y2).reshape(-1)
    out["g_eval"] = np.array(O.g_eval(PG, PD1, PD2, x, y2, z))
    for which, PD in (("y2", PD1), ("dem", PD2)):
        outs, grads, aux = O.critic_grads(PD, PG, y2, x, z, ep, which)
        out["critic_%s_outs" % which] = np.array(outs)
        out["critic_%s_gp" % which] = aux["gp"]
        out["critic_%s_gnorm" % which] = np.array(
            [float(np.sqrt((np.asarray(g, np.float64) ** 2).sum())) for g in grads.values()])
    outs, grads = O.g_grads(PG, PD1, PD2, x, y2, z)
    out["g_train_outs"] = np.array(outs)
    out["g_gnorm"] = np.array([float(np.sqrt((np.asarray(g, np.float64) ** 2).sum())) for g in grads.values()])
    tr = O.OracleTrainers(PG, PD1, PD2)
    s1 = tr.netD_y2_train([y2, x, z, ep])
    s2 = tr.netD_dem_train([y2, x, z, ep])
    s3 = tr.netG_train([x, y2, z])
    s4 = tr.netG_no_update([x, y2, z])
    out["seq_outs"] = np.array(s1 + s2 + s3 + s4)
    out["post_wsumG"], out["post_wsumD1"], out["post_wsumD2"] = checks(PG), checks(PD1), checks(PD2)
    np.savez_compressed(os.path.join(HERE, name + ".npz"), **out)
    print(name, "written")


def uresnet_case(name, img, B, seed, drop_seed):
    """DEP-UResNet: phase-0 predict, one phase-1 gradient evaluation (fp64 oracle) and two
    train_on_batch steps of the fp32 oracle."""
    import torch
    P = uresnet_params(seed)
    x, z, lab = O.synth_uresnet_batch(seed + 3, B, img, img)
    out = dict(img=img, B=B, seed=seed, drop_seed=drop_seed, wsum=checks(P),
               xsum=float(x.astype(np.float64).sum()), labsum=lab.reshape(-1, 4).sum(0))
    probs = O.uresnet_predict(P, x, z)
    rng = np.random.default_rng(321)
    idx = rng.integers(0, probs.size, 256)
    out["probs_idx"], out["probs_samples"] = idx, probs.reshape(-1)[idx]
    out["eval_loss"] = float(O.keras_categorical_crossentropy_t(torch.tensor(probs), torch.tensor(lab)))
    loss, grads, stats = O.uresnet_grads(P, x, z, lab, drop_seed=drop_seed, dtype=torch.float64)
    out["loss"] = loss
    out["gnorm"] = np.array([float(np.sqrt((np.asarray(g, np.float64) ** 2).sum())) for g in grads.values()])
    out["keep_sum"] = int(O.dropout_keep_mask(drop_seed, (B, img // 4, img // 4, 96)).sum())
    tr = O.OracleUResNet(P)
    out["step_losses"] = np.array([tr.train_on_batch([x, z], lab, drop_seed=drop_seed + k) for k in range(2)])
    out["post_wsum"] = checks(P)
    np.savez_compressed(os.path.join(HERE, name + ".npz"), **out)
    print(name, "written")


def uresnet_params(seed):
    """Random-init DEP-UResNet weights with a head small enough that the softmax is not saturated."""
    P = O.init_generator(seed, nc_out=4, randomize_bn=True, bias_std=0.05)
    P["gen_segmentation/kernel"] = (P["gen_segmentation/kernel"] * 0.05).astype(np.float32)
    return P


if __name__ == "__main__":
    case("small_64_b2", 64, 2, 11)
    uresnet_case("uresnet_64_b4", 64, 4, 41, 2024)
    if "--full" in sys.argv:
        case("full_256_b2", 256, 2, 21)
