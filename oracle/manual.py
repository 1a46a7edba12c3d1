"""Hand-derived backward restatement of the DEP-GAN train step (no autograd).

TEST INFRASTRUCTURE -- see ``oracle/__init__.py``.  PARITY UNPINNED.

This is the second, independent CPU implementation: every gradient here is
written out as the explicit sequence of convolutions / reductions that the HIP
path executes, including

  * the WGAN-GP double backward (A6): with ReLU/arg-max masks M_l constant a.e.,
        g_{l-1} = unpool( W_l^T * (M_l . g_l) ),  GP = mean_b (||g_0^b|| - 1)^2
        u_0 = dGP/dg_0,  v_l = W_l * u_{l-1},  u_l = M_l . v_l (gathered at the
        pool arg-max),  dGP/dW_l = corr(u_{l-1}, M_l . g_l),  no bias gradient;
  * phase-0 BatchNorm folded into the conv epilogue, with
        d gamma = rstd * ( sum_k W[k,co] dWraw[k,co] + (b - mu) * S[co] )
    so no second pass over the activations is needed (dWraw = un-scaled weight
    gradient, S = per-channel sum of the upstream gradient);
  * FiLM per-sample gradients and the noise-MLP backward.

``tests/test_oracle.py`` checks it against the autograd form in
``depgan_oracle.py`` (GT:523-598 of the reference training script).
torch is used only for conv/pool primitives; all tensors NCHW inside.

Mask-pinned evaluation.  Every function takes an optional ``masks`` argument: the
ReLU signs (GT:256-309, 319-335 ``Activation('relu')``), max-pool arg-maxes
(``MaxPooling2D``), FiLM-ReLU signs and the sign of the L1 term (GT:576) as data
instead of as functions of this evaluation's own values.  With the masks fixed the
step is multilinear in weights and inputs, so an fp32 evaluation (the HIP path)
and this fp64 one agree to rounding on EVERY input -- no unit can sit "on the
other side of its kink".  ``critic_masks`` / ``generator_masks`` build them from the
activations the HIP library hands out (``depgan_debug_tensor``), with the
library's own decision rules: ``act > 0``, first maximum of a 2x2 window in
row-major order, ``fl32(fl32(u * mul) + add) > 0`` for the FiLM blocks.
"""
from __future__ import annotations

import numpy as np
import torch
import torch.nn.functional as F

from .depgan_oracle import (BN_EPS, DIS_TRUNK, NOISE_HEADS, film_names, gen_trunk, to_torch,
                            trainable_names, _t)


def _es(eq, *ops):
    """einsum with float64 accumulation whatever the operands' type: when the convolutions of an evaluation run in
    float32 (the batch-32 256x256 test: a float64 step is minutes of CPU), the parameter-gradient reductions -- dot
    products over up to 6e6 pixel terms -- still sum in float64, so the oracle's own error stays at the 1e-6 of its
    convolutions instead of the 1e-4 of a long float32 sum."""
    return torch.einsum(eq, *[o.double() for o in ops]).to(ops[0].dtype)


def _rs(t, dims=None):
    """sum over dims with float64 accumulation (see _es)."""
    return (t.double().sum() if dims is None else t.double().sum(dims)).to(t.dtype)


def _es_batch(eq, a, b):
    """sum over the batch AND pixels of a (B,C,H,W) x (B,D,H,W) contraction (eq keeps `b`, e.g. "bchw,bdhw->bcd"): the
    pixel sums run per sample in the operands' type (one batched GEMM, 65536 terms each), the sum over samples in
    float64 -- casting whole activation tensors to float64 per filter tap cost the batch-32 test two minutes."""
    return torch.einsum(eq, a, b).double().sum(0).to(a.dtype)


def _rs_batch(t):
    """(B,C,H,W) -> (C,): pixel sums per sample in t's type, the sum over samples in float64."""
    return t.sum((2, 3)).double().sum(0).to(t.dtype)


def _w_oihw(w_hwio):
    return w_hwio.permute(3, 2, 0, 1)


def conv_fwd(x, w_hwio, b=None):
    return F.conv2d(x, _w_oihw(w_hwio), b, padding=w_hwio.shape[0] // 2)


def conv_bwd_data(dy, w_hwio):
    """dx = W^T * dy  ==  'same' conv of dy with the spatially flipped, io-transposed kernel."""
    wf = torch.flip(w_hwio, dims=(0, 1)).permute(2, 3, 0, 1)      # (Cin, Cout, kh, kw) as OIHW of the transposed conv
    return F.conv2d(dy, wf, None, padding=w_hwio.shape[0] // 2)


def conv_wgrad(x, dy, k):
    """dW[i,j,ci,co] = sum_{b,h,w} x[b,ci,h+i-p,w+j-p] dy[b,co,h,w]  (HWIO)."""
    p = k // 2
    xp = F.pad(x, (p, p, p, p))
    H, W = dy.shape[2], dy.shape[3]
    out = torch.empty((k, k, x.shape[1], dy.shape[1]), dtype=x.dtype)
    for i in range(k):
        for j in range(k):
            out[i, j] = _es_batch("bchw,bdhw->bcd", xp[:, :, i:i + H, j:j + W], dy)
    return out


def unpool(d, idx, shape):
    return F.max_unpool2d(d, idx, 2, output_size=shape)


def first_argmax_idx(a):
    """(B,C,H,W) array -> int64 (B,C,H/2,W/2) flat indices (into H*W) of the FIRST maximum of each 2x2 window in
    row-major order (0,0),(0,1),(1,0),(1,1): the rule of the HIP pooling kernels (csrc/ops.hip first_argmax4) and of
    torch's CPU max_pool2d."""
    a = np.asarray(a)
    B, C, H, W = a.shape
    w = a.reshape(B, C, H // 2, 2, W // 2, 2).transpose(0, 1, 2, 4, 3, 5).reshape(B, C, H // 2, W // 2, 4)
    k = np.argmax(w, axis=-1)                                  # first occurrence
    ii = np.arange(H // 2).reshape(1, 1, -1, 1)
    jj = np.arange(W // 2).reshape(1, 1, 1, -1)
    return ((2 * ii + k // 2) * W + 2 * jj + k % 2).astype(np.int64)


def _nchw_np(a):
    return np.ascontiguousarray(np.asarray(a).transpose(0, 3, 1, 2))


def critic_masks(acts_nhwc):
    """acts_nhwc: layer name -> post-ReLU activation (B,H,W,C) of one critic pass (any float array).
    Returns name -> (relu mask bool (B,C,H,W), pool arg-max indices or None)."""
    out = {}
    for name, k, ci, co, pool in DIS_TRUNK:
        a = _nchw_np(acts_nhwc[name])
        out[name] = (torch.from_numpy(a > 0), torch.from_numpy(first_argmax_idx(a)) if pool else None)
    return out


def generator_masks(outs_nhwc, us_nhwc, heads, a0, a1, attr=None, x=None, y2=None, nicg=1, fm=32):
    """Masks of one generator pass from the tensors the HIP library kept:
    outs_nhwc: trunk layer name -> output (post-ReLU; FiLM blocks not needed), us_nhwc: FiLM layer name -> BatchNorm
    output u, heads (B,1024) noise-head outputs in creation order, a0 / a1 (B,1024) noise-trunk activations.
    With attr, x, y2 (float32 NHWC) also the sign of the L1 term, evaluated as csrc/ops.hip g_dpre_kernel does:
    diff = attr - (y2 - x[..., 0]) in float32."""
    col, c0 = {}, 0
    for sfx, mult in NOISE_HEADS:
        col["noise_2_" + sfx] = (c0, c0 + fm * mult)
        c0 += fm * mult
    heads = np.asarray(heads, np.float32).reshape(-1, c0)
    m = {"noise_a0": torch.from_numpy(np.asarray(a0).reshape(-1, 32, fm) > 0),
         "noise_a1": torch.from_numpy(np.asarray(a1).reshape(-1, 32, fm) > 0)}
    for ent in gen_trunk(nicg, fm, 1):
        kind, name = ent[0], ent[1]
        if kind in ("conv", "deconv"):
            m[name] = torch.from_numpy(_nchw_np(outs_nhwc[name]) > 0)
        elif kind == "film":
            mul_n, add_n = film_names(ent[4])
            u = np.asarray(us_nhwc[name], np.float32)
            mul = heads[:, col[mul_n][0]:col[mul_n][1]][:, None, None, :]
            add = heads[:, col[add_n][0]:col[add_n][1]][:, None, None, :]
            v = (u * mul).astype(np.float32) + add               # two float32 roundings, as film_preact()
            m[name] = torch.from_numpy(_nchw_np(v > 0))
    for pool_name, conv_name in (("skip1", "gen_1"), ("skip2", "gen_3"), ("skip3", "gen_5")):
        m[pool_name] = torch.from_numpy(first_argmax_idx(_nchw_np(outs_nhwc[conv_name])))
    if attr is not None:
        at = np.asarray(attr, np.float32)
        diff = at - (np.asarray(y2, np.float32) - np.asarray(x, np.float32)[..., 0:1])
        m["sign"] = torch.from_numpy(_nchw_np(np.sign(diff)))
    return m


def gather_pool(u, idx):
    B, C = u.shape[:2]
    return u.reshape(B, C, -1).gather(2, idx.reshape(B, C, -1)).reshape(idx.shape)


# ----------------------------------------------------------------------------
# Critic
# ----------------------------------------------------------------------------
def d_forward_store(T, img_nchw, masks=None):
    acts = {"in": img_nchw}
    a = img_nchw
    for name, k, ci, co, pool in DIS_TRUNK:
        acts[name + "/x"] = a
        c = conv_fwd(a, T["conv2d_" + name + "/kernel"], T["conv2d_" + name + "/bias"])
        m = (c > 0) if masks is None else masks[name][0]
        a = c * m                                             # relu(c) under the evaluation's own mask
        acts[name], acts[name + "/m"] = a, m
        if pool:
            if masks is None:
                a, idx = F.max_pool2d(a, 2, return_indices=True)
            else:
                idx = masks[name][1]
                a = gather_pool(a, idx)
            acts[name + "/idx"] = idx
    w9 = T["dis_9/kernel"].reshape(-1)                       # (256,)
    wd = T["dense_1/kernel"].reshape(-1)                     # (hw,)
    B, C, h, w = a.shape
    t9 = (a * w9.view(1, C, 1, 1)).sum(1).reshape(B, h * w) + T["dis_9/bias"]
    out = t9 @ wd + T["dense_1/bias"]                        # (B,)
    acts["a11"], acts["t9"] = a, t9
    return out, acts


def d_backward(T, acts, c, want_wgrad=True, to_input=False):
    """Backward with per-sample upstream c (B,).  Returns (grads or None, dz dict, g_in)."""
    G = {}
    a11, t9 = acts["a11"], acts["t9"]
    B, C, h, w = a11.shape
    w9 = T["dis_9/kernel"].reshape(-1)
    wd = T["dense_1/kernel"].reshape(-1)
    if want_wgrad:
        G["dense_1/kernel"] = _rs(c[:, None] * t9, 0).reshape(-1, 1)
        G["dense_1/bias"] = c.sum().reshape(1)
        G["dis_9/kernel"] = _es("b,p,bcp->c", c, wd, a11.reshape(B, C, h * w)).reshape(1, 1, C, 1)
        G["dis_9/bias"] = (c.sum() * wd.sum()).reshape(1)
    d = c.view(B, 1, 1, 1) * w9.view(1, C, 1, 1) * wd.view(1, 1, h, w)
    dz = {}
    for li in range(len(DIS_TRUNK) - 1, -1, -1):
        name, k, ci, co, pool = DIS_TRUNK[li]
        if pool:
            d = unpool(d, acts[name + "/idx"], acts[name].shape[2:])
        d = d * acts[name + "/m"]
        dz[name] = d
        if want_wgrad:
            G["conv2d_" + name + "/kernel"] = conv_wgrad(acts[name + "/x"], d, k)
            G["conv2d_" + name + "/bias"] = _rs_batch(d)
        if li > 0 or to_input:
            d = conv_bwd_data(d, T["conv2d_" + name + "/kernel"])
    return (G if want_wgrad else None), dz, d


def d_gp_grads(T, acts, gz, g0, delta):
    """Weight gradient of delta * mean_b (||g0^b|| - 1)^2 (the double backward)."""
    B = g0.shape[0]
    norm = torch.sqrt((g0 ** 2).sum((1, 2, 3)))
    gp = ((norm - 1.0) ** 2).mean()
    u = (delta * 2.0 / B) * ((norm - 1.0) / norm).view(B, 1, 1, 1) * g0
    G = {}
    for name, k, ci, co, pool in DIS_TRUNK:
        G["conv2d_" + name + "/kernel"] = conv_wgrad(u, gz[name], k)
        v = conv_fwd(u, T["conv2d_" + name + "/kernel"])
        u = v * acts[name + "/m"]
        if pool:
            u = gather_pool(u, acts[name + "/idx"])
    Bc, C, h, w = u.shape
    w9 = T["dis_9/kernel"].reshape(-1)
    wd = T["dense_1/kernel"].reshape(-1)
    G["dis_9/kernel"] = _es("bcp,p->c", u.reshape(Bc, C, h * w), wd).reshape(1, 1, C, 1)
    G["dense_1/kernel"] = _es("bcp,c->p", u.reshape(Bc, C, h * w), w9).reshape(-1, 1)
    return G, gp, norm


def critic_grads_manual(PD, real, fake, ep, delta=10.0, dtype=torch.float64, masks=None):
    """real/fake: (B,H,W,1) NHWC numpy.  Returns ([loss_real, loss_fake], grads dict, aux).
    masks: None, or the three passes' masks (real, fake, mixed), each as critic_masks() returns them."""
    mr, mf, mm = masks if masks is not None else (None, None, None)
    T = to_torch(PD, dtype)
    r = _t(real, dtype).permute(0, 3, 1, 2)
    f = _t(fake, dtype).permute(0, 3, 1, 2)
    e = _t(ep, dtype).reshape(-1, 1, 1, 1)
    B = r.shape[0]
    mixed = e * r + (1.0 - e) * f
    out_r, acts_r = d_forward_store(T, r, mr)
    out_f, acts_f = d_forward_store(T, f, mf)
    out_m, acts_m = d_forward_store(T, mixed, mm)
    G_r, _, _ = d_backward(T, acts_r, torch.full((B,), -1.0 / B, dtype=dtype))
    G_f, _, _ = d_backward(T, acts_f, torch.full((B,), 1.0 / B, dtype=dtype))
    _, gz, g0 = d_backward(T, acts_m, torch.ones(B, dtype=dtype), want_wgrad=False, to_input=True)
    G_gp, gp, norm = d_gp_grads(T, acts_m, gz, g0, delta)
    grads = {}
    for n in trainable_names(PD):
        g = G_r[n] + G_f[n]
        if n in G_gp:
            g = g + G_gp[n]
        grads[n] = g.numpy()
    aux = dict(gp=float(gp), norm=norm.numpy(), g0=g0.permute(0, 2, 3, 1).numpy(),
               decisions=tuple({n: (a[n + "/m"], a.get(n + "/idx")) for n, *_ in DIS_TRUNK} for a in (acts_r, acts_f, acts_m)))
    return [float(out_r.mean()), float(out_f.mean())], grads, aux


def d_input_grad(PD, img, dtype=torch.float64, masks=None):
    """g0 = d sum_b D(img)_b / d img   (B,H,W,1)."""
    T = to_torch(PD, dtype)
    x = _t(img, dtype).permute(0, 3, 1, 2)
    out, acts = d_forward_store(T, x, masks)
    _, _, g0 = d_backward(T, acts, torch.ones(x.shape[0], dtype=dtype), want_wgrad=False, to_input=True)
    return out.numpy(), g0.permute(0, 2, 3, 1).numpy()


# ----------------------------------------------------------------------------
# Generator
# ----------------------------------------------------------------------------
def _bn_st(T, name):
    rstd = torch.rsqrt(T[name + "/moving_variance"] + BN_EPS)
    s = T[name + "/gamma"] * rstd
    t = T[name + "/beta"] - T[name + "/moving_mean"] * s
    return s, t, rstd


def noise_fwd_store(T, z, masks=None):
    st = {}
    s0, t0, _ = _bn_st(T, "dense_bn_noise_1_add_f0")
    h0 = z @ T["dense_noise_1_add_f0/kernel"] + T["dense_noise_1_add_f0/bias"]     # (B,32,fm)
    p0 = h0 * s0 + t0
    m0 = (p0 > 0) if masks is None else masks["noise_a0"]
    a0 = p0 * m0
    s1, t1, _ = _bn_st(T, "dense_bn_noise_1_add_f1")
    h1 = a0 @ T["dense_noise_1_add_f1/kernel"] + T["dense_noise_1_add_f1/bias"]
    p1 = h1 * s1 + t1
    m1 = (p1 > 0) if masks is None else masks["noise_a1"]
    a1 = p1 * m1
    flat = a1.reshape(a1.shape[0], -1)
    heads = {}
    for sfx, _ in NOISE_HEADS:
        n = "noise_2_" + sfx
        s, t, _ = _bn_st(T, "dense_bn_" + n)
        heads[n] = (flat @ T["dense_" + n + "/kernel"] + T["dense_" + n + "/bias"]) * s + t
    st.update(z=z, h0=h0, a0=a0, h1=h1, a1=a1, flat=flat, m0=m0, m1=m1)
    return heads, st


def noise_bwd(T, st, dheads):
    G = {}
    flat = st["flat"]
    dflat = torch.zeros_like(flat)
    for sfx, _ in NOISE_HEADS:
        n = "noise_2_" + sfx
        dh = dheads[n]                                          # (B,C) grad at BN output
        s, t, rstd = _bn_st(T, "dense_bn_" + n)
        lin = flat @ T["dense_" + n + "/kernel"] + T["dense_" + n + "/bias"]
        G["dense_bn_" + n + "/beta"] = dh.sum(0)
        G["dense_bn_" + n + "/gamma"] = (dh * (lin - T["dense_bn_" + n + "/moving_mean"]) * rstd).sum(0)
        dl = dh * s
        G["dense_" + n + "/kernel"] = flat.t() @ dl
        G["dense_" + n + "/bias"] = dl.sum(0)
        dflat = dflat + dl @ T["dense_" + n + "/kernel"].t()
    da1 = dflat.reshape(st["a1"].shape)
    for nm, hin, h, am in (("noise_1_add_f1", st["a0"], st["h1"], st["m1"]),
                           ("noise_1_add_f0", st["z"], st["h0"], st["m0"])):
        s, t, rstd = _bn_st(T, "dense_bn_" + nm)
        dy = da1 * am
        G["dense_bn_" + nm + "/beta"] = dy.sum((0, 1))
        G["dense_bn_" + nm + "/gamma"] = (dy * (h - T["dense_bn_" + nm + "/moving_mean"]) * rstd).sum((0, 1))
        dl = dy * s
        G["dense_" + nm + "/kernel"] = torch.einsum("bpi,bpo->io", hin, dl)
        G["dense_" + nm + "/bias"] = dl.sum((0, 1))
        da1 = dl @ T["dense_" + nm + "/kernel"].t()
    return G


def g_forward_store(T, x_nchw, z, nicg=1, fm=32, masks=None):
    heads, nst = noise_fwd_store(T, z, masks)
    st = {"noise": nst, "heads": heads}
    a = x_nchw
    skips = {}
    for ent in gen_trunk(nicg, fm, 1):
        kind, name = ent[0], ent[1]
        if kind == "conv":
            s, t, _ = _bn_st(T, "bn_" + name)
            st[name + "/x"] = a
            c = conv_fwd(a, T["conv2d_" + name + "/kernel"], T["conv2d_" + name + "/bias"])
            p = c * s.view(1, -1, 1, 1) + t.view(1, -1, 1, 1)
            m = (p > 0) if masks is None else masks[name]
            a = p * m
            st[name], st[name + "/m"] = a, m
        elif kind == "film":
            mul_n, add_n = film_names(ent[4])
            s, t, _ = _bn_st(T, "bn_" + name)
            st[name + "/x"] = a
            c = conv_fwd(a, T["conv2d_" + name + "/kernel"], T["conv2d_" + name + "/bias"])
            u = c * s.view(1, -1, 1, 1) + t.view(1, -1, 1, 1)
            v = u * heads[mul_n][:, :, None, None] + heads[add_n][:, :, None, None]
            m = (v > 0) if masks is None else masks[name]
            st[name + "/u"], st[name + "/vpos"] = u, m
            a = v * m + a
            st[name] = a
        elif kind == "pool":
            skips[name] = a
            st[name + "/shape"] = a.shape[2:]
            if masks is None:
                a, idx = F.max_pool2d(a, 2, return_indices=True)
            else:
                idx = masks[name]
                a = gather_pool(a, idx)
            st[name + "/idx"] = idx
        elif kind == "deconv":
            s, t, _ = _bn_st(T, "bn_" + name)
            st[name + "/x"] = a
            w = T["deconv2d_" + name + "/kernel"]
            c = F.conv_transpose2d(a, w.permute(3, 2, 0, 1), T["deconv2d_" + name + "/bias"], stride=2)
            p = c * s.view(1, -1, 1, 1) + t.view(1, -1, 1, 1)
            m = (p > 0) if masks is None else masks[name]
            o = p * m
            st[name], st[name + "/m"] = o, m
            a = torch.cat([o, skips[ent[4]]], dim=1)
        elif kind == "head":
            st[name + "/x"] = a
            pre = conv_fwd(a, T[name + "/kernel"], T[name + "/bias"])
            a = torch.tanh(pre)
            st[name] = a
    return a, st


def _conv_bn_bwd(T, G, name, x, dy, k=3):
    """dy = grad at the BN output.  Fills G for conv+bn `name`; returns dx."""
    s, t, rstd = _bn_st(T, "bn_" + name)
    W = T["conv2d_" + name + "/kernel"]
    b = T["conv2d_" + name + "/bias"]
    S = _rs_batch(dy)
    dWraw = conv_wgrad(x, dy, k)
    G["bn_" + name + "/beta"] = S
    G["bn_" + name + "/gamma"] = rstd * ((W * dWraw).sum((0, 1, 2)) + (b - T["bn_" + name + "/moving_mean"]) * S)
    G["conv2d_" + name + "/kernel"] = dWraw * s
    G["conv2d_" + name + "/bias"] = S * s
    return conv_bwd_data(dy * s.view(1, -1, 1, 1), W)


def g_backward(T, st, d_attr, nicg=1, fm=32):
    """d_attr: grad wrt attribution (B,1,H,W).  Returns grads dict (torch)."""
    G = {}
    heads = st["heads"]
    dheads = {}
    skip_grad = {}
    d = None
    trunk = gen_trunk(nicg, fm, 1)
    for ent in reversed(trunk):
        kind, name = ent[0], ent[1]
        if kind == "head":
            a = st[name]
            x = st[name + "/x"]
            dpre = d_attr * (1.0 - a * a)
            G[name + "/kernel"] = _es_batch("bchw,bohw->bco", x, dpre).reshape(1, 1, x.shape[1], 1)
            G[name + "/bias"] = _rs_batch(dpre).reshape(1)
            d = dpre * T[name + "/kernel"].reshape(1, -1, 1, 1)
        elif kind == "conv":
            dy = d * st[name + "/m"]
            d = _conv_bn_bwd(T, G, name, st[name + "/x"], dy)
        elif kind == "film":
            mul_n, add_n = film_names(ent[4])
            dv = d * st[name + "/vpos"]
            dheads[add_n] = dv.sum((2, 3))
            dheads[mul_n] = (dv * st[name + "/u"]).sum((2, 3))
            du = dv * heads[mul_n][:, :, None, None]
            d = _conv_bn_bwd(T, G, name, st[name + "/x"], du) + d
        elif kind == "pool":
            d = unpool(d, st[name + "/idx"], st[name + "/shape"]) + skip_grad[name]
        elif kind == "deconv":
            co = ent[3]
            s, t, rstd = _bn_st(T, "bn_" + name)
            skip_grad[ent[4]] = d[:, co:]
            do = d[:, :co] * st[name + "/m"]
            x = st[name + "/x"]
            W = T["deconv2d_" + name + "/kernel"]                 # (2,2,Co,Ci)
            S = _rs_batch(do)
            dWraw = torch.empty_like(W)
            for di in range(2):
                for dj in range(2):
                    dWraw[di, dj] = _es_batch("bohw,bihw->boi", do[:, :, di::2, dj::2], x)
            G["bn_" + name + "/beta"] = S
            G["bn_" + name + "/gamma"] = rstd * ((W * dWraw).sum((0, 1, 3)) +
                                                  (T["deconv2d_" + name + "/bias"] - T["bn_" + name + "/moving_mean"]) * S)
            G["deconv2d_" + name + "/kernel"] = dWraw * s.view(1, 1, -1, 1)
            G["deconv2d_" + name + "/bias"] = S * s
            dos = do * s.view(1, -1, 1, 1)
            dx = torch.zeros_like(x)
            for di in range(2):
                for dj in range(2):
                    dx = dx + torch.einsum("bohw,oi->bihw", dos[:, :, di::2, dj::2], W[di, dj])
            d = dx
    G.update(noise_bwd(T, st["noise"], dheads))
    return G


def g_grads_manual(PG, PDy2, PDdem, x, y2, z, thr=0.5, nicg=1, dtype=torch.float64, masks=None, decisions=None):
    """Manual netG_train gradients (GT:574-594).  Returns (6 scalars, grads).
    masks: None, or (generator_masks(...), critic_masks of D_y2(fake_y2), critic_masks of D_dem(attr)).
    decisions: an empty dict that receives the masks this evaluation used (same three-part form)."""
    mg, md1, md2 = masks if masks is not None else (None, None, None)
    TG = to_torch(PG, dtype)
    TD1 = to_torch(PDy2, dtype)
    TD2 = to_torch(PDdem, dtype)
    xt = _t(x, dtype).permute(0, 3, 1, 2)
    y2t = _t(y2, dtype).permute(0, 3, 1, 2)
    zt = _t(z, dtype)
    B, _, H, W = xt.shape
    y1 = xt[:, 0:1]
    real_dem = y2t - y1
    attr, st = g_forward_store(TG, xt, zt, nicg, masks=mg)
    fake_y2 = y1 + attr
    o1, acts1 = d_forward_store(TD1, fake_y2, md1)
    o2, acts2 = d_forward_store(TD2, attr, md2)
    ones = torch.ones(B, dtype=dtype)
    _, _, g1 = d_backward(TD1, acts1, ones, want_wgrad=False, to_input=True)
    _, _, g2 = d_backward(TD2, acts2, ones, want_wgrad=False, to_input=True)
    diff = attr - real_dem
    sgn = torch.sign(diff) if (mg is None or "sign" not in mg) else mg["sign"].to(dtype)
    d_attr = -(g1 + g2) / B + (100.0 / diff.numel()) * sgn
    G = g_backward(TG, st, d_attr, nicg)
    loss_fake, loss_fake_dem = o1.mean(), o2.mean()
    m1 = diff.abs().mean() * 100.0
    wr = (y2t >= thr).to(dtype)
    wf = (fake_y2 >= thr).to(dtype)
    dice = (2.0 * (wr * wf).sum() + 1e-7) / (wr.sum() + wf.sum() + 1e-7)
    m4 = 1.0 - dice
    m3 = ((wr.sum() / 1000.0 - wf.sum() / 1000.0) ** 2) * 100.0
    loss = -loss_fake - loss_fake_dem + m1 + m3 + m4
    grads = {n: G[n].numpy() for n in trainable_names(PG)}
    if decisions is not None:
        dg = {"noise_a0": st["noise"]["m0"], "noise_a1": st["noise"]["m1"], "sign": sgn}
        for ent in gen_trunk(nicg, 32, 1):
            kind, name = ent[0], ent[1]
            if kind in ("conv", "deconv"):
                dg[name] = st[name + "/m"]
            elif kind == "film":
                dg[name] = st[name + "/vpos"]
            elif kind == "pool":
                dg[name] = st[name + "/idx"]
        decisions["G"] = dg
        decisions["D_y2"] = {n: (acts1[n + "/m"], acts1.get(n + "/idx")) for n, *_ in DIS_TRUNK}
        decisions["D_dem"] = {n: (acts2[n + "/m"], acts2.get(n + "/idx")) for n, *_ in DIS_TRUNK}
    return [float(v) for v in (loss, loss_fake, loss_fake_dem, m1, m3, m4)], grads


def count_decision_flips(a, b):
    """Number of differing entries (ReLU signs, arg-maxes, L1 signs) between two decision sets as critic_masks() /
    generator_masks() / the `decisions` outputs above hold them, and the number of decisions compared."""
    flips = total = 0
    for k in a:
        va, vb = a[k], b[k]
        for ta, tb in (zip(va, vb) if isinstance(va, tuple) else ((va, vb),)):
            if ta is None:
                continue
            flips += int((torch.as_tensor(ta).to(torch.float64) != torch.as_tensor(tb).to(torch.float64)).sum())
            total += int(torch.as_tensor(ta).numel())
    return flips, total
