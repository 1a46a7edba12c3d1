"""CPU restatement of the DEP-GAN two-critic WGAN-GP training step (autograd form).

TEST INFRASTRUCTURE -- see ``oracle/__init__.py``.  PARITY UNPINNED (no
reference golden vectors exist; Keras/TF cannot run here).

Follows /root/reference/DEP-GAN_PROB_IM_twoCritics_training_4fold.py ("GT"):
  * layer helpers            GT:254-312
  * critic  Dis_C2D_FCN1     GT:316-345
  * generator Gen_UNet2D     GT:349-498
  * losses / updates / the four K.function closures   GT:523-598
and the Keras-2.x/TF-1.x default semantics listed in SURVEY.md Appendix B
(channels_last, HWIO kernels, Conv2DTranspose kernel (kh,kw,Cout,Cin), BN
eps=1e-3 in inference mode because no closure feeds K.learning_phase(),
Dropout identity in phase 0, Keras Adam with epsilon 1e-7).

All tensors at this module's boundary are NumPy, NHWC, Keras layouts.  torch
(CPU, fp32 by default) is used as the array/autograd engine only.
"""
from __future__ import annotations

import math
from collections import OrderedDict

import numpy as np
import torch
import torch.nn.functional as F

BN_EPS = 1e-3          # keras BatchNormalization default epsilon (App. B.3)
ADAM_EPS = 1e-7        # K.epsilon() (App. B.6)
NOISE_SIZE = 32        # GT:41

# ----------------------------------------------------------------------------
# Layer tables (GT:316-345, GT:349-498)
# ----------------------------------------------------------------------------
# film key -> suffix used in the noise-head layer names (GT:363-395)
NOISE_HEADS = [  # creation order, (suffix, width multiplier)
    ("add_m3", 3), ("mul_m3", 3), ("add_m2", 2), ("mul_m2", 2),
    ("add_m1", 1), ("mul_m1", 1), ("add", 4), ("mul", 4),
    ("add_p3", 3), ("mul_p3", 3), ("add_p2", 2), ("mul_p2", 2),
    ("add_p1", 1), ("mul_p1", 1),
]

# Generator trunk, in forward order.  kind: conv (conv+bn+relu), film
# (conv+bn, FiLM, relu, + residual), pool, deconv (2x2 s2 + bn + relu, then
# concat with the named skip), head (1x1 conv + tanh/softmax).
def gen_trunk(nicg=1, fm=32, nc_out=1):
    f1, f2, f3, f4 = fm, 2 * fm, 3 * fm, 4 * fm
    return [
        ("conv", "gen_0", nicg, f1), ("film", "gen_noise_m1", f1, f1, "m1"),
        ("conv", "gen_1", f1, f1), ("pool", "skip1"),
        ("conv", "gen_2", f1, f2), ("film", "gen_noise_m2", f2, f2, "m2"),
        ("conv", "gen_3", f2, f2), ("pool", "skip2"),
        ("conv", "gen_4", f2, f3), ("film", "gen_noise_m3", f3, f3, "m3"),
        ("conv", "gen_5", f3, f3), ("pool", "skip3"),
        ("conv", "gen_8", f3, f4), ("film", "gen_noise_p4", f4, f4, ""),
        ("conv", "gen_9", f4, f4),
        ("deconv", "de_gen_9", f4, f4, "skip3"),
        ("conv", "gen_10", f4 + f3, f3), ("film", "gen_noise_p3", f3, f3, "p3"),
        ("conv", "gen_11", f3, f3),
        ("deconv", "de_gen_11", f3, f3, "skip2"),
        ("conv", "gen_14", f3 + f2, f2), ("film", "gen_noise_p2", f2, f2, "p2"),
        ("conv", "gen_15", f2, f2),
        ("deconv", "de_gen_15", f2, f2, "skip1"),
        ("conv", "gen_16", f2 + f1, f1), ("film", "gen_noise_p1", f1, f1, "p1"),
        ("conv", "gen_17", f1, f1),
        ("head", "gen_segmentation", f1, nc_out),
    ]

# Critic trunk (GT:319-339): (name, k, cin, cout, pool_after)
DIS_TRUNK = [
    ("dis_0a", 5, 1, 16, False), ("dis_0b", 5, 16, 16, True),
    ("dis_1a", 5, 16, 32, False), ("dis_1b", 5, 32, 32, True),
    ("dis_2", 3, 32, 64, False), ("dis_3", 3, 64, 64, True),
    ("dis_4", 3, 64, 128, False), ("dis_5", 3, 128, 128, True),
    ("dis_6", 3, 128, 256, False), ("dis_7", 3, 256, 256, False),
    ("dis_8", 3, 256, 256, False),
]


def film_names(key):
    sfx = ("_" + key) if key else ""
    return "noise_2_mul" + sfx, "noise_2_add" + sfx


# ----------------------------------------------------------------------------
# Parameter construction (Keras layouts; App. B.1, B.7)
# ----------------------------------------------------------------------------
def _glorot_uniform(rng, shape, fan_in, fan_out):
    lim = math.sqrt(6.0 / (fan_in + fan_out))
    return rng.uniform(-lim, lim, size=shape).astype(np.float32)


def _he_normal(rng, shape, fan_in):
    # keras he_normal = truncated normal(+-2 sigma), stddev sqrt(2/fan_in)/.8796
    std = math.sqrt(2.0 / fan_in) / 0.87962566103423978
    v = rng.standard_normal(size=shape)
    bad = np.abs(v) > 2
    while bad.any():
        v[bad] = rng.standard_normal(size=int(bad.sum()))
        bad = np.abs(v) > 2
    return (v * std).astype(np.float32)


def _bn(P, name, c, rng, randomize):
    if randomize:  # make phase-0 BN a non-trivial affine
        P[name + "/gamma"] = rng.uniform(0.5, 1.5, c).astype(np.float32)
        P[name + "/beta"] = (0.1 * rng.standard_normal(c)).astype(np.float32)
        P[name + "/moving_mean"] = (0.1 * rng.standard_normal(c)).astype(np.float32)
        P[name + "/moving_variance"] = rng.uniform(0.5, 1.5, c).astype(np.float32)
    else:
        P[name + "/gamma"] = np.ones(c, np.float32)
        P[name + "/beta"] = np.zeros(c, np.float32)
        P[name + "/moving_mean"] = np.zeros(c, np.float32)
        P[name + "/moving_variance"] = np.ones(c, np.float32)


def init_generator(seed, nicg=1, fm=32, nc_out=1, randomize_bn=True, bias_std=0.0):
    """Parameters of Gen_UNet2D((256,256,nicg),(32,1),fm,nc_out), GT:349-498.
    Canonical order: noise MLP (f0, f1, 14 heads in creation order) then the
    trunk in forward order.  Keys are '<keras layer name>/<weight name>'."""
    rng = np.random.default_rng(seed)
    P = OrderedDict()

    def dense(name, fin, fout):
        P["dense_" + name + "/kernel"] = _he_normal(rng, (fin, fout), fin)
        P["dense_" + name + "/bias"] = (bias_std * rng.standard_normal(fout)).astype(np.float32)
        _bn(P, "dense_bn_" + name, fout, rng, randomize_bn)

    dense("noise_1_add_f0", 1, fm)           # GT:358
    dense("noise_1_add_f1", fm, fm)          # GT:359
    for sfx, mult in NOISE_HEADS:            # GT:363-395
        dense("noise_2_" + sfx, NOISE_SIZE * fm, fm * mult)
    for ent in gen_trunk(nicg, fm, nc_out):
        kind, name = ent[0], ent[1]
        if kind in ("conv", "film"):
            ci, co = ent[2], ent[3]
            P["conv2d_" + name + "/kernel"] = _glorot_uniform(rng, (3, 3, ci, co), 9 * ci, 9 * co)
            P["conv2d_" + name + "/bias"] = (bias_std * rng.standard_normal(co)).astype(np.float32)
            _bn(P, "bn_" + name, co, rng, randomize_bn)
        elif kind == "deconv":
            ci, co = ent[2], ent[3]
            # Conv2DTranspose kernel layout (kh, kw, Cout, Cin)  (App. B.1)
            P["deconv2d_" + name + "/kernel"] = _glorot_uniform(rng, (2, 2, co, ci), 4 * co, 4 * ci)
            P["deconv2d_" + name + "/bias"] = (bias_std * rng.standard_normal(co)).astype(np.float32)
            _bn(P, "bn_" + name, co, rng, randomize_bn)
        elif kind == "head":
            ci, co = ent[2], ent[3]
            P[name + "/kernel"] = _glorot_uniform(rng, (1, 1, ci, co), ci, co)
            P[name + "/bias"] = (bias_std * rng.standard_normal(co)).astype(np.float32)
    return P


def init_critic(seed, bias_std=0.0, img=256):
    """Parameters of Dis_C2D_FCN1((img,img,1)), GT:316-345 (img=256 in the
    reference; smaller img only for fast tests: Flatten is (img/16)^2 long)."""
    rng = np.random.default_rng(seed)
    P = OrderedDict()
    for name, k, ci, co, _ in DIS_TRUNK:
        P["conv2d_" + name + "/kernel"] = _glorot_uniform(rng, (k, k, ci, co), k * k * ci, k * k * co)
        P["conv2d_" + name + "/bias"] = (bias_std * rng.standard_normal(co)).astype(np.float32)
    P["dis_9/kernel"] = _he_normal(rng, (1, 1, 256, 1), 256)          # GT:339
    P["dis_9/bias"] = (bias_std * rng.standard_normal(1)).astype(np.float32)
    nflat = (img // 16) ** 2
    P["dense_1/kernel"] = _he_normal(rng, (nflat, 1), nflat)          # GT:342
    P["dense_1/bias"] = (bias_std * rng.standard_normal(1)).astype(np.float32)
    return P


def trainable_names(P):
    return [k for k in P if not (k.endswith("/moving_mean") or k.endswith("/moving_variance"))]


# ----------------------------------------------------------------------------
# torch helpers
# ----------------------------------------------------------------------------
def to_torch(P, dtype=torch.float32, requires_grad=False):
    T = OrderedDict()
    for k, v in P.items():
        t = torch.tensor(np.asarray(v), dtype=dtype)
        if requires_grad and not (k.endswith("/moving_mean") or k.endswith("/moving_variance")):
            t.requires_grad_(True)
        T[k] = t
    return T


def _t(a, dtype):
    return a if isinstance(a, torch.Tensor) else torch.tensor(np.asarray(a), dtype=dtype)


def _nchw(x):
    return x.permute(0, 3, 1, 2)


def _nhwc(x):
    return x.permute(0, 2, 3, 1)


# BASELINE config 4 on the bf16 matrix pipe: "the same graph with bf16-rounded operands".  When the switch is
# on, the activation operand of every convolution that the HIP build runs on v_mfma_f32_32x32x16_bf16 (Cout a multiple
# of 32, Cin >= 8 and a multiple of 4 -- csrc/igemm_bf16.hip, dg_plan_conv_bf16) is rounded to bf16 (RNE) first; the
# weights are rounded by round_kernels_bf16.  Straight-through in the backward pass (the HIP build additionally rounds
# the gradient operand of its backward-data convolutions: the gradient tolerance of config 4 covers that).
_ACT_BF16 = False


class bf16_activations:
    def __init__(self, on=True):
        self.on = on

    def __enter__(self):
        global _ACT_BF16
        self.prev, _ACT_BF16 = _ACT_BF16, self.on

    def __exit__(self, *exc):
        global _ACT_BF16
        _ACT_BF16 = self.prev


def _act_operand(x, cin, cout):
    if not _ACT_BF16 or cout % 32 != 0 or cin < 8 or cin % 4 != 0:
        return x
    q = x.detach().to(torch.float32).to(torch.bfloat16).to(x.dtype)
    return x + (q - x.detach())


def _conv_same(x, w_hwio, b):
    """keras Conv2D(padding='same', stride 1), NCHW in/out, HWIO kernel."""
    k = w_hwio.shape[0]
    x = _act_operand(x, w_hwio.shape[2], w_hwio.shape[3])
    return F.conv2d(x, w_hwio.permute(3, 2, 0, 1), b, padding=k // 2)


def _bn_infer(x, T, name, ch_axis=1):
    """phase-0 BN: tf.nn.batch_normalization form  x*inv + (beta - mean*inv)."""
    inv = T[name + "/gamma"] * torch.rsqrt(T[name + "/moving_variance"] + BN_EPS)
    sh = T[name + "/beta"] - T[name + "/moving_mean"] * inv
    shape = [1] * x.dim()
    shape[ch_axis] = -1
    return x * inv.view(shape) + sh.view(shape)


# ----------------------------------------------------------------------------
# Generator forward (phase 0)  GT:349-498
# ----------------------------------------------------------------------------
def noise_mlp(T, z):
    """z (B,32,1) -> dict head-name -> (B, C).  GT:358-395, App. B.5."""
    h = z @ T["dense_noise_1_add_f0/kernel"] + T["dense_noise_1_add_f0/bias"]     # (B,32,fm)
    h = torch.relu(_bn_infer(h, T, "dense_bn_noise_1_add_f0", ch_axis=2))
    h = h @ T["dense_noise_1_add_f1/kernel"] + T["dense_noise_1_add_f1/bias"]
    h = torch.relu(_bn_infer(h, T, "dense_bn_noise_1_add_f1", ch_axis=2))
    flat = h.reshape(h.shape[0], -1)                                              # (B,1024) (pos,feat)
    heads = {}
    for sfx, _ in NOISE_HEADS:
        n = "noise_2_" + sfx
        v = flat @ T["dense_" + n + "/kernel"] + T["dense_" + n + "/bias"]
        heads[n] = _bn_infer(v, T, "dense_bn_" + n, ch_axis=1)
    return heads


def g_forward_t(T, x, z, nicg=1, fm=32, nc_out=1, head="tanh", taps=None):
    """x (B,H,W,nicg) NHWC torch, z (B,32,1) -> (B,H,W,nc_out) NHWC torch."""
    heads = noise_mlp(T, z)
    a = _nchw(x)
    skips = {}
    for ent in gen_trunk(nicg, fm, nc_out):
        kind, name = ent[0], ent[1]
        if kind == "conv":
            a = _conv_same(a, T["conv2d_" + name + "/kernel"], T["conv2d_" + name + "/bias"])
            a = torch.relu(_bn_infer(a, T, "bn_" + name))
        elif kind == "film":
            mul_n, add_n = film_names(ent[4])
            u = _conv_same(a, T["conv2d_" + name + "/kernel"], T["conv2d_" + name + "/bias"])
            u = _bn_infer(u, T, "bn_" + name)
            v = u * heads[mul_n][:, :, None, None] + heads[add_n][:, :, None, None]
            a = torch.relu(v) + a
        elif kind == "pool":
            skips[name] = a
            a = F.max_pool2d(a, 2)
        elif kind == "deconv":
            w = T["deconv2d_" + name + "/kernel"]            # (kh,kw,Cout,Cin)
            a = _act_operand(a, w.shape[3], w.shape[2])
            a = F.conv_transpose2d(a, w.permute(3, 2, 0, 1), T["deconv2d_" + name + "/bias"], stride=2)
            a = torch.relu(_bn_infer(a, T, "bn_" + name))
            a = torch.cat([a, skips[ent[4]]], dim=1)         # GT:450 order [deconv, skip]
        elif kind == "head":
            a = _conv_same(a, T[name + "/kernel"], T[name + "/bias"])
            if head == "tanh":
                a = torch.tanh(a)
            elif head == "softmax":
                a = torch.softmax(a, dim=1)
        if taps is not None:
            taps[name] = a
    return _nhwc(a)


def d_forward_t(T, img, taps=None):
    """img (B,H,W,1) NHWC torch -> (B,1).  GT:316-345."""
    a = _nchw(img)
    for name, k, ci, co, pool in DIS_TRUNK:
        a = torch.relu(_conv_same(a, T["conv2d_" + name + "/kernel"], T["conv2d_" + name + "/bias"]))
        if taps is not None:
            taps[name] = a
        if pool:
            a = F.max_pool2d(a, 2)
    a = _conv_same(a, T["dis_9/kernel"], T["dis_9/bias"])       # (B,1,h,w)
    flat = _nhwc(a).reshape(a.shape[0], -1)                     # Flatten of NHWC
    return flat @ T["dense_1/kernel"] + T["dense_1/bias"]


# numpy-facing wrappers -------------------------------------------------------
def g_predict(P, x, z, nicg=1, fm=32, nc_out=1, head="tanh", dtype=torch.float32):
    with torch.no_grad():
        T = to_torch(P, dtype)
        return g_forward_t(T, _t(x, dtype), _t(z, dtype), nicg, fm, nc_out, head).numpy()


def d_predict(P, img, dtype=torch.float32):
    with torch.no_grad():
        T = to_torch(P, dtype)
        return d_forward_t(T, _t(img, dtype)).numpy()


# ----------------------------------------------------------------------------
# Keras Adam (App. B.6)
# ----------------------------------------------------------------------------
class KerasAdam:
    def __init__(self, names, lr, beta_1=0.0, beta_2=0.9, eps=ADAM_EPS):
        self.lr, self.b1, self.b2, self.eps = lr, beta_1, beta_2, eps
        self.iterations = 0
        self.names = list(names)
        self.m = {}
        self.v = {}

    def apply(self, P, grads):
        """P: dict name->np array (updated in place); grads: dict name->np array.
        beta_1 / beta_2 are float32 backend variables in Keras, so `1. - self.beta_2` is a float32 subtraction
        (1 - 0.999f, not float32(0.001)); the moments follow in float32."""
        t = self.iterations + 1
        lr_t = self.lr * math.sqrt(1.0 - self.b2 ** t) / (1.0 - self.b1 ** t)
        for n in self.names:
            dt = P[n].dtype.type
            b1, b2, one = dt(self.b1), dt(self.b2), dt(1.0)
            g = np.asarray(grads[n], dtype=P[n].dtype)
            m = self.m.get(n, np.zeros_like(P[n]))
            v = self.v.get(n, np.zeros_like(P[n]))
            m = b1 * m + (one - b1) * g
            v = b2 * v + (one - b2) * g * g
            P[n] = (P[n] - dt(lr_t) * m / (np.sqrt(v) + dt(self.eps))).astype(P[n].dtype)
            self.m[n], self.v[n] = m, v
        self.iterations = t


# ----------------------------------------------------------------------------
# Loss graphs  GT:523-598
# ----------------------------------------------------------------------------
def _critic_loss_t(TD, real, fake, ep, delta):
    """WGAN-GP critic loss on (real, fake) images.  GT:536-547 / GT:555-566.
    Returns (loss, loss_real, loss_fake, grad_penalty, norm per sample)."""
    mixed = ep * real + (1.0 - ep) * fake
    mixed = mixed.detach().requires_grad_(True)
    loss_real = d_forward_t(TD, real).mean()
    loss_fake = d_forward_t(TD, fake).mean()
    out_mixed = d_forward_t(TD, mixed)
    (grad_mixed,) = torch.autograd.grad(out_mixed.sum(), mixed, create_graph=True)
    norm = torch.sqrt((grad_mixed ** 2).sum(dim=(1, 2, 3)))
    gp = ((norm - 1.0) ** 2).mean()
    loss = loss_fake - loss_real + delta * gp
    return loss, loss_real, loss_fake, gp, norm, grad_mixed


def critic_grads(PD, PG, y2, x, z, ep, which, delta=10.0, nicg=1, dtype=torch.float32):
    """Gradients of the critic loss w.r.t. the critic's trainable weights.
    which='y2' -> GT:533-552 ; which='dem' -> GT:555-571.
    inputs in the reference order [y2, x, z, ep].  Returns (outs[2], grads, aux)."""
    TD = to_torch(PD, dtype, requires_grad=True)
    TG = to_torch(PG, dtype)
    y2_t, x_t, z_t, ep_t = _t(y2, dtype), _t(x, dtype), _t(z, dtype), _t(ep, dtype)
    y1 = x_t[..., 0:1]                                           # GT:528-529
    with torch.no_grad():
        attr = g_forward_t(TG, x_t, z_t, nicg=nicg)              # GT:533 (no grad into G here)
    if which == "y2":
        real, fake = y2_t, y1 + attr                             # GT:534
    else:
        real, fake = y2_t - y1, attr                             # GT:530, 557
    loss, lr_, lf_, gp, norm, gmix = _critic_loss_t(TD, real, fake, ep_t.reshape(-1, 1, 1, 1), delta)
    names = trainable_names(PD)
    gs = torch.autograd.grad(loss, [TD[n] for n in names], allow_unused=True)
    grads = {n: (g.detach().numpy() if g is not None else np.zeros_like(PD[n])) for n, g in zip(names, gs)}
    aux = dict(loss=float(loss.detach()), gp=float(gp.detach()), norm=norm.detach().numpy(),
               grad_mixed=gmix.detach().numpy(), attr=attr.numpy())
    return [float(lr_.detach()), float(lf_.detach())], grads, aux


def _g_loss_t(TG, TDy2, TDdem, x_t, y2_t, z_t, thr, nicg):
    """Generator loss, GT:574-592.  Returns the 6 reported scalars (torch)."""
    y1 = x_t[..., 0:1]
    real_dem = y2_t - y1                                         # GT:530
    attr = g_forward_t(TG, x_t, z_t, nicg=nicg)                  # GT:533
    fake_y2 = y1 + attr                                          # GT:534
    loss_fake = d_forward_t(TDy2, fake_y2).mean()                # GT:541
    loss_fake_dem = d_forward_t(TDdem, attr).mean()              # GT:560
    m1 = (attr - real_dem).abs().mean() * 100.0                  # GT:576
    wr = (y2_t >= thr).to(y2_t.dtype)                            # GT:581 (no gradient)
    wf = (fake_y2.detach() >= thr).to(y2_t.dtype)                # GT:582 (no gradient)
    inter = (wr * wf).sum()
    dice = (2.0 * inter + 1e-7) / (wr.sum() + wf.sum() + 1e-7)   # GT:153-157
    m4 = (1.0 - dice) * 1.0                                      # GT:583
    m3 = ((wr.sum() / 1000.0 - wf.sum() / 1000.0) ** 2) * 100.0  # GT:587-589
    loss = (-loss_fake) + (-loss_fake_dem) + m1 + m3 + m4        # GT:592
    return loss, loss_fake, loss_fake_dem, m1, m3, m4


def g_eval(PG, PDy2, PDdem, x, y2, z, thr=0.5, nicg=1, dtype=torch.float32):
    """netG_no_update([x, y2, z]) -> 6 scalars.  GT:595-596."""
    with torch.no_grad():
        outs = _g_loss_t(to_torch(PG, dtype), to_torch(PDy2, dtype), to_torch(PDdem, dtype),
                         _t(x, dtype), _t(y2, dtype), _t(z, dtype), thr, nicg)
    return [float(o) for o in outs]


def g_grads(PG, PDy2, PDdem, x, y2, z, thr=0.5, nicg=1, dtype=torch.float32):
    """Gradient of the generator loss w.r.t. netG.trainable_weights.  GT:594."""
    TG = to_torch(PG, dtype, requires_grad=True)
    outs = _g_loss_t(TG, to_torch(PDy2, dtype), to_torch(PDdem, dtype),
                     _t(x, dtype), _t(y2, dtype), _t(z, dtype), thr, nicg)
    names = trainable_names(PG)
    gs = torch.autograd.grad(outs[0], [TG[n] for n in names], allow_unused=True)
    grads = {n: (g.detach().numpy() if g is not None else np.zeros_like(PG[n])) for n, g in zip(names, gs)}
    return [float(o.detach()) for o in outs], grads


# ----------------------------------------------------------------------------
# The four closures as one stateful object (GT:549-598)
# ----------------------------------------------------------------------------
def round_bf16(a):
    """float32 -> nearest bfloat16 (ties to even) -> float32, on the bit pattern."""
    u = np.ascontiguousarray(a, np.float32).view(np.uint32).astype(np.uint64)
    r = ((u + 0x7FFF + ((u >> 16) & 1)) & 0xFFFF0000).astype(np.uint32)
    return r.view(np.float32).reshape(np.shape(a))


def round_kernels_bf16(P):
    """BASELINE config 4 ("bf16 weights with fp32 accumulate"): the compute copy of a weight dict -- every
    '/kernel' tensor rounded to bf16, biases / BN parameters / moving statistics untouched."""
    return OrderedDict((k, round_bf16(v) if k.endswith("/kernel") else v) for k, v in P.items())


class OracleTrainers:
    """netD_y2_train / netD_dem_train / netG_no_update / netG_train with the
    reference's positional contracts; weights are NumPy dicts updated in place.
    weights_dtype="bfloat16": gradients are taken at the bf16-rounded kernels and applied to the fp32 masters."""

    def __init__(self, PG, PDy2, PDdem, lrD=1e-4, lrG=1e-4, delta=10.0, thr=0.5, nicg=1,
                 dtype=torch.float32, weights_dtype="float32", activations_dtype="float32"):
        self._PG, self._PDy2, self._PDdem = PG, PDy2, PDdem
        self._act = activations_dtype == "bfloat16"
        self._q = round_kernels_bf16 if weights_dtype == "bfloat16" else (lambda P: P)
        self.delta, self.thr, self.nicg, self.dtype = delta, thr, nicg, dtype
        self.optD_y2 = KerasAdam(trainable_names(PDy2), lrD, 0.0, 0.9)   # GT:549
        self.optD_dem = KerasAdam(trainable_names(PDdem), lrD, 0.0, 0.9)  # GT:568
        self.optG = KerasAdam(trainable_names(PG), lrG, 0.0, 0.9)        # GT:594

    def netD_y2_train(self, inputs):
        with bf16_activations(self._act):
            return self._netD_y2_train(inputs)

    def _netD_y2_train(self, inputs):
        y2, x, z, ep = inputs
        outs, grads, _ = critic_grads(self.PDy2, self.PG, y2, x, z, ep, "y2", self.delta, self.nicg, self.dtype)
        self.optD_y2.apply(self._PDy2, grads)
        return outs

    def netD_dem_train(self, inputs):
        with bf16_activations(self._act):
            return self._netD_dem_train(inputs)

    def _netD_dem_train(self, inputs):
        y2, x, z, ep = inputs
        outs, grads, _ = critic_grads(self.PDdem, self.PG, y2, x, z, ep, "dem", self.delta, self.nicg, self.dtype)
        self.optD_dem.apply(self._PDdem, grads)
        return outs

    def netG_no_update(self, inputs):
        with bf16_activations(self._act):
            return self._netG_no_update(inputs)

    def _netG_no_update(self, inputs):
        x, y2, z = inputs
        return g_eval(self.PG, self.PDy2, self.PDdem, x, y2, z, self.thr, self.nicg, self.dtype)

    def netG_train(self, inputs):
        with bf16_activations(self._act):
            return self._netG_train(inputs)

    def _netG_train(self, inputs):
        x, y2, z = inputs
        outs, grads = g_grads(self.PG, self.PDy2, self.PDdem, x, y2, z, self.thr, self.nicg, self.dtype)
        self.optG.apply(self._PG, grads)
        return outs

    # the weights the graph reads: the masters, or their bf16-rounded compute copies
    @property
    def PG(self):
        return self._q(self._PG)

    @property
    def PDy2(self):
        return self._q(self._PDy2)

    @property
    def PDdem(self):
        return self._q(self._PDdem)


# ----------------------------------------------------------------------------
# Synthetic inputs
# ----------------------------------------------------------------------------
def synth_batch(seed, B, H=256, W=256, nicg=1):
    """Seeded synthetic (x, y2, z, ep): sparse blob maps in [0,1] inside an
    elliptical 'brain' mask (mirrors GT:685-687, 715-716), z~N(0,1) (GT:807),
    ep~U[0,1) (GT:808; both drawn in float64 then cast like the reference)."""
    rng = np.random.default_rng(seed)
    yy, xx = np.mgrid[0:H, 0:W].astype(np.float64)
    cy, cx = (H - 1) / 2.0 + 0.5, (W - 1) / 2.0 + 0.5
    mask = (((yy - cy) / (100.0 * H / 256)) ** 2 + ((xx - cx) / (80.0 * W / 256)) ** 2) <= 1.0
    x0 = np.zeros((B, H, W), np.float64)
    y2 = np.zeros((B, H, W), np.float64)
    sc = H / 256.0
    for b in range(B):
        K = int(rng.integers(3, 9))
        cys = cy + rng.uniform(-70, 70, K) * sc
        cxs = cx + rng.uniform(-55, 55, K) * sc
        amp = rng.uniform(0.3, 1.0, K)
        sig = rng.uniform(2.0, 8.0, K) * sc
        amp2 = amp * rng.uniform(0.7, 1.3, K)
        sig2 = sig * rng.uniform(0.7, 1.3, K)
        keep = rng.uniform(size=K) > 0.15
        for k in range(K):
            d2 = (yy - cys[k]) ** 2 + (xx - cxs[k]) ** 2
            x0[b] += amp[k] * np.exp(-d2 / (2 * sig[k] ** 2))
            if keep[k]:
                y2[b] += amp2[k] * np.exp(-d2 / (2 * sig2[k] ** 2))
        x0[b] = np.clip(x0[b] + 0.15 * rng.uniform(size=(H, W)) ** 4, 0, 1) * mask
        y2[b] = np.clip(y2[b] + 0.15 * rng.uniform(size=(H, W)) ** 4, 0, 1) * mask
    x = x0[..., None]
    if nicg == 2:
        fl = mask * (0.35 + 0.4 * x0 + 0.05 * rng.standard_normal((B, H, W)))
        fl = (fl - fl.min()) / max(fl.max() - fl.min(), 1e-12)
        x = np.stack([x0, fl], axis=-1)
    z = rng.normal(size=(B, NOISE_SIZE, 1))
    ep = rng.uniform(size=(B, 1, 1, 1))
    return (x.astype(np.float32), y2[..., None].astype(np.float32),
            z.astype(np.float32), ep.astype(np.float32))


# ----------------------------------------------------------------------------
# DEP-UResNet supervised path (reference
# /root/reference/DEP-UResNet-wNoises-training-4fold.py, "UT")
#   same Gen_UNet2D with ONE Dropout(0.25) after conv_10 (UT:388), nc_out = 4 + softmax (UT:424),
#   compile(Adam(1e-4), 'categorical_crossentropy') (UT:427), trained with Model.fit (UT:602-606):
#   learning phase 1 => batch-statistics BatchNorm + moving-average updates, active Dropout.
# ----------------------------------------------------------------------------
BN_MOMENTUM = 0.99     # keras default (UT:26 defines bn_momentum but never passes it)
DROP_RATE = 0.25


def hash_uniform_u32(seed, n):
    """Counter-based RNG shared bit-for-bit with the HIP kernel (murmur3 finaliser of index*golden ^ seed)."""
    i = np.arange(n, dtype=np.uint64)
    x = ((i * np.uint64(0x9E3779B1)) & np.uint64(0xFFFFFFFF)) ^ np.uint64(seed & 0xFFFFFFFF)
    x ^= x >> np.uint64(16)
    x = (x * np.uint64(0x85EBCA6B)) & np.uint64(0xFFFFFFFF)
    x ^= x >> np.uint64(13)
    x = (x * np.uint64(0xC2B2AE35)) & np.uint64(0xFFFFFFFF)
    x ^= x >> np.uint64(16)
    return x.astype(np.uint32)


def dropout_keep_mask(seed, shape_nhwc, rate=DROP_RATE):
    """keep[b,h,w,c] in {0,1}: element dropped when hash < rate * 2^32 (NHWC linear index)."""
    n = int(np.prod(shape_nhwc))
    thr = np.uint32(int(rate * 4294967296.0))
    return (hash_uniform_u32(seed, n) >= thr).reshape(shape_nhwc)


def _bn_train(x, T, name, ch_axis, stats):
    """phase-1 BN: batch mean / biased variance over all non-channel axes (App. B.3)."""
    axes = [a for a in range(x.dim()) if a != ch_axis]
    mean = x.mean(dim=axes)
    var = x.var(dim=axes, unbiased=False)
    shape = [1] * x.dim()
    shape[ch_axis] = -1
    y = (x - mean.view(shape)) * torch.rsqrt(var.view(shape) + BN_EPS) * T[name + "/gamma"].view(shape) \
        + T[name + "/beta"].view(shape)
    if stats is not None:
        n = x.numel() // x.shape[ch_axis]
        stats[name] = (mean.detach(), var.detach(), n, x.dim() == 4)
    return y


def _pool_gather(a, idx):
    B, C = a.shape[:2]
    return a.reshape(B, C, -1).gather(2, idx.reshape(B, C, -1)).reshape(idx.shape)


def uresnet_forward_t(T, x, z, phase=1, keep_mask=None, stats=None, fm=32, nc_out=4, masks=None):
    """DEP-UResNet forward.  x (B,H,W,1), z (B,32,1); keep_mask: NHWC {0,1} mask for do_gen_1 (phase 1).
    Returns softmax probabilities (B,H,W,nc_out).
    masks: None, or the ReLU signs / pool arg-maxes as data (oracle.manual.generator_masks: what the HIP path decided) --
    relu(v) becomes v * mask and max-pooling a gather, which makes the step smooth in the weights (tests/test_gpu_masked.py
    explains why the gradient tests pin them)."""
    def bn(v, name, ch_axis=1):
        return _bn_train(v, T, name, ch_axis, stats) if phase == 1 else _bn_infer(v, T, name, ch_axis)

    def relu(v, key):
        return torch.relu(v) if masks is None else v * masks[key]

    h = z @ T["dense_noise_1_add_f0/kernel"] + T["dense_noise_1_add_f0/bias"]
    h = relu(bn(h, "dense_bn_noise_1_add_f0", 2), "noise_a0")
    h = h @ T["dense_noise_1_add_f1/kernel"] + T["dense_noise_1_add_f1/bias"]
    h = relu(bn(h, "dense_bn_noise_1_add_f1", 2), "noise_a1")
    flat = h.reshape(h.shape[0], -1)
    heads = {}
    for sfx, _ in NOISE_HEADS:
        n = "noise_2_" + sfx
        heads[n] = bn(flat @ T["dense_" + n + "/kernel"] + T["dense_" + n + "/bias"], "dense_bn_" + n, 1)
    a = _nchw(x)
    skips = {}
    for ent in gen_trunk(1, fm, nc_out):
        kind, name = ent[0], ent[1]
        if kind == "conv":
            a = relu(bn(_conv_same(a, T["conv2d_" + name + "/kernel"], T["conv2d_" + name + "/bias"]), "bn_" + name), name)
            if name == "gen_10" and phase == 1 and keep_mask is not None:      # do_gen_1, UT:388
                a = a * _nchw(keep_mask.to(a.dtype)) / (1.0 - DROP_RATE)
        elif kind == "film":
            mul_n, add_n = film_names(ent[4])
            u = bn(_conv_same(a, T["conv2d_" + name + "/kernel"], T["conv2d_" + name + "/bias"]), "bn_" + name)
            a = relu(u * heads[mul_n][:, :, None, None] + heads[add_n][:, :, None, None], name) + a
        elif kind == "pool":
            skips[name] = a
            a = F.max_pool2d(a, 2) if masks is None else _pool_gather(a, masks[name])
        elif kind == "deconv":
            w = T["deconv2d_" + name + "/kernel"]
            a = F.conv_transpose2d(a, w.permute(3, 2, 0, 1), T["deconv2d_" + name + "/bias"], stride=2)
            a = relu(bn(a, "bn_" + name), name)
            a = torch.cat([a, skips[ent[4]]], dim=1)
        elif kind == "head":
            a = torch.softmax(_conv_same(a, T[name + "/kernel"], T[name + "/bias"]), dim=1)
    return _nhwc(a)


def keras_categorical_crossentropy_t(p, t):
    """keras.losses.categorical_crossentropy on probabilities (App. B.9): renormalise, clip, -sum t log p, mean."""
    p = p / p.sum(dim=-1, keepdim=True)
    p = torch.clamp(p, 1e-7, 1.0 - 1e-7)
    return (-(t * torch.log(p)).sum(dim=-1)).mean()


def uresnet_predict(P, x, z, dtype=torch.float32):
    with torch.no_grad():
        return uresnet_forward_t(to_torch(P, dtype), _t(x, dtype), _t(z, dtype), phase=0).numpy()


def uresnet_grads(P, x, z, labels, drop_seed=None, dtype=torch.float32, masks=None):
    """One Model.train_on_batch worth of gradients (phase 1).  labels: one-hot (B,H,W,4).
    Returns (loss, grads dict, batch BN stats dict).  masks: see uresnet_forward_t."""
    T = to_torch(P, dtype, requires_grad=True)
    xt, zt, lt = _t(x, dtype), _t(z, dtype), _t(np.asarray(labels, np.float32), dtype)
    keep = None
    if drop_seed is not None:
        B, H, W, _ = xt.shape
        keep = torch.tensor(dropout_keep_mask(drop_seed, (B, H // 4, W // 4, 96)))
    stats = {}
    p = uresnet_forward_t(T, xt, zt, phase=1, keep_mask=keep, stats=stats, masks=masks)
    loss = keras_categorical_crossentropy_t(p, lt)
    names = trainable_names(P)
    gs = torch.autograd.grad(loss, [T[n] for n in names], allow_unused=True)
    grads = {n: (g.detach().numpy() if g is not None else np.zeros_like(P[n])) for n, g in zip(names, gs)}
    return float(loss.detach()), grads, stats


class OracleUResNet:
    """my_network.train_on_batch / fit step (UT:427, 602-606): Adam(1e-4, .9, .999) + BN moving averages."""

    def __init__(self, P, lr=1e-4, dtype=torch.float32):
        self.P, self.dtype = P, dtype
        self.opt = KerasAdam(trainable_names(P), lr, 0.9, 0.999)

    def train_on_batch(self, inputs, labels, drop_seed=None, masks=None):
        x, z = inputs
        loss, grads, stats = uresnet_grads(self.P, x, z, labels, drop_seed, self.dtype, masks)
        self.opt.apply(self.P, grads)
        for name, (mean, var, n, fused) in stats.items():
            # moving variance: Bessel-corrected on the fused 4-D path, n/(n-(1+eps)) on the generic path
            corr = n / (n - 1.0) if fused else n / (n - (1.0 + BN_EPS))
            mm, mv = self.P[name + "/moving_mean"], self.P[name + "/moving_variance"]
            self.P[name + "/moving_mean"] = (mm * BN_MOMENTUM + mean.numpy() * (1 - BN_MOMENTUM)).astype(mm.dtype)
            self.P[name + "/moving_variance"] = (mv * BN_MOMENTUM + var.numpy() * corr * (1 - BN_MOMENTUM)).astype(mv.dtype)
        return loss

    def predict(self, inputs):
        return uresnet_predict(self.P, inputs[0], inputs[1], self.dtype)


def synth_uresnet_batch(seed, B, H=256, W=256, thr=0.178):
    """z-scored FLAIR-like slice, labels {0,1,2,3} from thresholding x0 / y2
    (1 shrink, 2 grow, 3 stay; coding per GE:722-741), one-hot (B,H,W,4) (UT:565-568)."""
    x0, y2, z, _ = synth_batch(seed, B, H, W)
    rng = np.random.default_rng(seed + 1)
    flair = 0.35 + 0.4 * x0 + 0.05 * rng.standard_normal(x0.shape)
    flair = ((flair - flair.mean()) / flair.std()).astype(np.float32)          # UT:511 z-score
    a, b = x0[..., 0] >= thr, y2[..., 0] >= thr
    lab = np.zeros(a.shape, np.int64)
    lab[a & ~b] = 1
    lab[~a & b] = 2
    lab[a & b] = 3
    onehot = np.eye(4, dtype=np.float32)[lab]
    return flair, z, onehot
