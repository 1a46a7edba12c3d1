"""GPU parity, operator level, through the C ABI: every HIP convolution form
(MFMA implicit GEMM, direct edge kernels, MFMA / VALU weight gradient) against a
float64 torch-CPU reference of the same op, including ragged spatial sizes,
channel counts that exercise every kernel variant and the concat/strided views."""
import ctypes as C

import numpy as np
import pytest
import torch
import torch.nn.functional as F

pytestmark = pytest.mark.gpu
TOL = 1e-4   # fp32 MFMA == fmaf chain; only summation order differs from the reference


def P(t):
    return C.c_void_p(t.data_ptr())


def rel(a, b):
    a, b = np.asarray(a, np.float64), np.asarray(b, np.float64)
    return float(np.abs(a - b).max() / (np.abs(b).max() + 1e-30))


def _ref_conv(x, w, b, relu):
    y = F.conv2d(torch.from_numpy(x).permute(0, 3, 1, 2).double(), torch.from_numpy(w).permute(3, 2, 0, 1).double(),
                 None if b is None else torch.from_numpy(b).double(), padding=w.shape[0] // 2)
    return (torch.relu(y) if relu else y).permute(0, 2, 3, 1).numpy()


CONV_CASES = [  # B,H,W,Cin,Cout,k,path
    (2, 32, 32, 32, 32, 3, 1), (2, 48, 40, 32, 64, 3, 1), (1, 32, 32, 224, 96, 3, 1), (2, 21, 19, 64, 160, 3, 1),
    (2, 32, 32, 16, 16, 5, 1), (2, 32, 32, 16, 32, 5, 1), (2, 32, 32, 32, 32, 5, 1), (2, 17, 33, 32, 16, 5, 1),
    (2, 32, 32, 128, 128, 1, 1), (2, 32, 32, 48, 48, 3, 1), (1, 16, 16, 256, 256, 3, 1),
    # path 6: the 8-channel-chunk form of the 3x3 MF = 32 kernel (what launches of >= 1536 items take)
    (2, 32, 32, 32, 32, 3, 6), (2, 48, 40, 32, 64, 3, 6), (1, 32, 32, 224, 96, 3, 6), (2, 21, 19, 64, 160, 3, 6),
    (1, 16, 16, 256, 256, 3, 6), (2, 30, 18, 8, 32, 3, 6),
    # path 7: the wave-private form (no workgroup barrier in steady state; Cin <= 64): ragged sizes, one and several
    # channel tiles, one to eight K chunks, persistent (>= 8 super-tiles per XCD form) and one-item workgroups
    (2, 32, 32, 32, 32, 3, 7), (2, 48, 40, 32, 64, 3, 7), (2, 21, 19, 64, 160, 3, 7), (2, 30, 18, 8, 32, 3, 7),
    (8, 64, 128, 64, 64, 3, 7), (3, 16, 64, 16, 96, 3, 7),
    # path 8: Winograd F(2x2,3x3) on the fp32 matrix pipe (igemm_wino.hip; even H and W, Cin % 8, Cout % 32): ragged
    # 16 x 16 tiles, one to 28 K chunks, several channel tiles, one-item and persistent workgroups
    (2, 32, 32, 32, 32, 3, 8), (2, 48, 40, 32, 64, 3, 8), (1, 32, 32, 224, 96, 3, 8), (2, 22, 18, 64, 160, 3, 8),
    (1, 16, 16, 256, 256, 3, 8), (2, 30, 18, 8, 32, 3, 8), (32, 64, 64, 32, 64, 3, 8), (5, 100, 72, 40, 32, 3, 8),
    (2, 32, 32, 1, 32, 3, 2), (2, 32, 32, 2, 32, 3, 2), (2, 30, 18, 1, 16, 5, 2), (2, 32, 32, 16, 1, 5, 2),
    # single output channel (dD/dx): the 4-pixels-per-thread kernel, ragged tiles, channel tails, both kernel sizes
    (2, 45, 70, 16, 1, 5, 2), (2, 33, 31, 8, 1, 3, 2), (1, 40, 40, 6, 1, 5, 2), (1, 20, 36, 12, 1, 3, 2),
    # one or two input channels: the 4-pixels x 4-channels-per-thread kernel
    (1, 37, 50, 2, 16, 5, 2), (1, 20, 20, 1, 32, 5, 2), (2, 19, 33, 2, 16, 3, 2),
]


@pytest.mark.parametrize("case", CONV_CASES)
def test_conv_forward_and_backward_data(lib, case):
    from dep_gan_im_amd import _lib
    B, H, W, ci, co, k, path = case
    dev = torch.device("cuda:0")
    rng = np.random.default_rng(ci * 1000 + co + k)
    x = rng.standard_normal((B, H, W, ci)).astype(np.float32)
    w = (rng.standard_normal((k, k, ci, co)) / np.sqrt(k * k * ci)).astype(np.float32)
    b = rng.standard_normal(co).astype(np.float32)
    dy = rng.standard_normal((B, H, W, co)).astype(np.float32)
    xd, wd, bd, dyd = [torch.from_numpy(a).to(dev) for a in (x, w, b, dy)]
    out = torch.full((B, H, W, co), float("nan"), device=dev)
    _lib.check(lib.depgan_op_conv2d(P(xd), P(wd), P(bd), P(out), B, H, W, ci, co, k, 1, path, None))
    torch.cuda.synchronize()
    assert rel(out.cpu().numpy(), _ref_conv(x, w, b, True)) < TOL
    dx = torch.full((B, H, W, ci), float("nan"), device=dev)
    # path 6 (8-channel chunks) exists for 32-channel output tiles only: the backward of a layer with fewer input
    # channels than that runs the 16-channel-chunk kernel
    bpath = 1 if (path in (6, 7, 8) and ci % 32) else path
    if bpath == 7 and co > 64:
        bpath = 6            # the backward of this layer has more than 64 input channels: workgroup tiles
    _lib.check(lib.depgan_op_conv2d_bwd_data(P(dyd), P(wd), P(dx), B, H, W, ci, co, k, bpath, None))
    torch.cuda.synchronize()
    xt = torch.from_numpy(x).permute(0, 3, 1, 2).double().requires_grad_(True)
    y = F.conv2d(xt, torch.from_numpy(w).permute(3, 2, 0, 1).double(), padding=k // 2)
    (gx,) = torch.autograd.grad(y, xt, torch.from_numpy(dy).permute(0, 3, 1, 2).double())
    assert rel(dx.cpu().numpy(), gx.permute(0, 2, 3, 1).numpy()) < TOL


@pytest.mark.parametrize("case", [(8, 128, 128, 32, 32), (8, 128, 128, 64, 64), (32, 64, 64, 32, 64), (4, 100, 72, 40, 32)])
def test_wave_private_conv_is_bit_identical_to_the_tile_kernel(lib, case):
    """csrc/igemm_wp.hip walks K in the order of igemm_conv_kernel<32,3,8,9> (chunk -> tap -> four MFMAs) on the same
    packed panel and shares its epilogue text: forward and backward-data results must be the same bits, on launches
    large enough for its persistent form (every workgroup several items, one weight panel each)."""
    from dep_gan_im_amd import _lib
    B, H, W, ci, co = case
    dev = torch.device("cuda:0")
    g = torch.Generator(device="cpu").manual_seed(ci + co + H)
    x = torch.randn((B, H, W, ci), generator=g).to(dev)
    w = (torch.randn((3, 3, ci, co), generator=g) / (3.0 * ci ** 0.5)).to(dev)
    b = torch.randn((co,), generator=g).to(dev)
    outs = []
    for path in (6, 7):
        out = torch.full((B, H, W, co), float("nan"), device=dev)
        _lib.check(lib.depgan_op_conv2d(P(x), P(w), P(b), P(out), B, H, W, ci, co, 3, 1, path, None))
        outs.append(out)
    torch.cuda.synchronize()
    assert torch.equal(outs[0], outs[1])
    if ci % 32 == 0 and co <= 64:
        dy = torch.randn((B, H, W, co), generator=g).to(dev)
        dxs = []
        for path in (6, 7):
            dx = torch.full((B, H, W, ci), float("nan"), device=dev)
            _lib.check(lib.depgan_op_conv2d_bwd_data(P(dy), P(w), P(dx), B, H, W, ci, co, 3, path, None))
            dxs.append(dx)
        torch.cuda.synchronize()
        assert torch.equal(dxs[0], dxs[1])


def _bf16(a):
    """float32 -> nearest bf16 (ties to even) -> float32, what v_cvt_pk_bf16_f32 does to both MFMA operands."""
    return torch.from_numpy(np.ascontiguousarray(a, np.float32)).to(torch.bfloat16).to(torch.float32).numpy()


BF16_CASES = [  # B,H,W,Cin,Cout,k: every variant of igemm_bf16.hip, ragged tiles, Cin below / not a multiple of a chunk
    (2, 32, 32, 32, 32, 3), (2, 48, 40, 32, 64, 3), (1, 32, 32, 224, 96, 3), (2, 21, 19, 64, 160, 3),
    (2, 32, 32, 16, 32, 5), (2, 32, 32, 32, 32, 5), (2, 17, 33, 32, 64, 5), (2, 32, 32, 128, 128, 1),
    (2, 32, 32, 48, 96, 3), (1, 16, 16, 256, 256, 3), (2, 30, 18, 8, 32, 3), (2, 32, 32, 384, 96, 1),
    # launches of a thousand and more workgroups; ragged tiles and a channel tail among them
    (9, 112, 120, 32, 64, 3), (6, 96, 104, 40, 32, 5), (16, 64, 64, 64, 96, 1),
]


@pytest.mark.parametrize("case", BF16_CASES)
def test_bf16_mfma_conv_forward_and_backward_data(lib, case):
    """BASELINE configs[3] on the bf16 matrix pipe (v_mfma_f32_32x32x16_bf16): both operands rounded to bf16 (RNE), fp32
    accumulation, fp32 epilogue.  The reference multiplies the SAME rounded operands in float64, so what is left is
    the summation order: the fp32 tolerance holds."""
    from dep_gan_im_amd import _lib
    B, H, W, ci, co, k = case
    dev = torch.device("cuda:0")
    rng = np.random.default_rng(ci * 1000 + co + k)
    x = rng.standard_normal((B, H, W, ci)).astype(np.float32)
    w = (rng.standard_normal((k, k, ci, co)) / np.sqrt(k * k * ci)).astype(np.float32)
    b = rng.standard_normal(co).astype(np.float32)
    dy = rng.standard_normal((B, H, W, co)).astype(np.float32)
    xd, wd, bd, dyd = [torch.from_numpy(a).to(dev) for a in (x, w, b, dy)]
    out = torch.full((B, H, W, co), float("nan"), device=dev)
    _lib.check(lib.depgan_op_conv2d(P(xd), P(wd), P(bd), P(out), B, H, W, ci, co, k, 1, 3, None))
    torch.cuda.synchronize()
    got = out.cpu().numpy()
    assert rel(got, _ref_conv(_bf16(x), _bf16(w), b, True)) < TOL
    assert rel(got, _ref_conv(x, w, b, True)) > 5 * TOL           # the rounding is there (and we follow it)
    if ci % 32 == 0 and ci >= 8 and co >= 8 and co % 4 == 0:       # backward-data = a convolution with Cout_bwd = Cin
        dx = torch.full((B, H, W, ci), float("nan"), device=dev)
        _lib.check(lib.depgan_op_conv2d_bwd_data(P(dyd), P(wd), P(dx), B, H, W, ci, co, k, 3, None))
        torch.cuda.synchronize()
        xt = torch.from_numpy(x).permute(0, 3, 1, 2).double().requires_grad_(True)
        y = F.conv2d(xt, torch.from_numpy(_bf16(w)).permute(3, 2, 0, 1).double(), padding=k // 2)
        (gx,) = torch.autograd.grad(y, xt, torch.from_numpy(_bf16(dy)).permute(0, 3, 1, 2).double())
        assert rel(dx.cpu().numpy(), gx.permute(0, 2, 3, 1).numpy()) < TOL
    # shapes the bf16 kernel does not cover are refused on this path (the model falls back to the fp32 pipe for them)
    assert lib.depgan_op_conv2d(P(xd), P(wd), P(bd), P(out), B, H, W, ci, 16, k, 1, 3, None) != 0   # refused before any launch


SPLIT_CASES = [(2, 32, 32, 32, 32, 3), (2, 48, 40, 32, 64, 3), (1, 32, 32, 224, 96, 3), (2, 21, 19, 64, 160, 3),
               (2, 32, 32, 16, 32, 5), (2, 17, 33, 32, 64, 5), (2, 32, 32, 128, 128, 1), (1, 16, 16, 256, 256, 3),
               (2, 30, 18, 8, 32, 3), (2, 32, 32, 48, 96, 3), (2, 32, 32, 16, 16, 5), (2, 17, 33, 32, 16, 5), (2, 32, 32, 32, 48, 3)]


@pytest.mark.parametrize("case", SPLIT_CASES)
def test_split_bf16_conv_is_fp32_grade(lib, case):
    """depgan_config.f32_split (opt-in): fp32 operands split exactly into bf16 terms, the largest cross products on the
    bf16 matrix pipe, fp32 accumulation.  Against the float64 reference of the UNROUNDED operands: six products must be
    as accurate as the native fp32 MFMA kernel (same tolerance, errors printed side by side), three products within
    2^-16-sized terms."""
    from dep_gan_im_amd import _lib
    B, H, W, ci, co, k = case
    dev = torch.device("cuda:0")
    rng = np.random.default_rng(ci * 1000 + co + k)
    x = rng.standard_normal((B, H, W, ci)).astype(np.float32)
    w = (rng.standard_normal((k, k, ci, co)) / np.sqrt(k * k * ci)).astype(np.float32)
    b = rng.standard_normal(co).astype(np.float32)
    dy = rng.standard_normal((B, H, W, co)).astype(np.float32)
    xd, wd, bd, dyd = [torch.from_numpy(a).to(dev) for a in (x, w, b, dy)]
    ref = _ref_conv(x, w, b, True)
    errs = {}
    for name, path in (("native", 1), ("x3", 4), ("x6", 5)):
        out = torch.full((B, H, W, co), float("nan"), device=dev)
        _lib.check(lib.depgan_op_conv2d(P(xd), P(wd), P(bd), P(out), B, H, W, ci, co, k, 1, path, None))
        torch.cuda.synchronize()
        got = out.cpu().numpy().astype(np.float64)
        errs[name] = (rel(got, ref), float(np.abs(got - ref).mean() / np.abs(ref).mean()))
    print("k%d %d->%d: max-rel / mean-rel error vs fp64: native %.1e / %.1e, six products %.1e / %.1e, three %.1e / %.1e"
          % ((k, ci, co) + errs["native"] + errs["x6"] + errs["x3"]))
    assert errs["native"][0] < TOL and errs["x6"][0] < TOL
    assert errs["x6"][1] < 4 * errs["native"][1] + 1e-7           # fp32-grade: the same order as the native pipe
    assert errs["x3"][0] < 3e-4 and errs["x3"][1] < 2e-5
    if ci % 16 == 0:
        xt = torch.from_numpy(x).permute(0, 3, 1, 2).double().requires_grad_(True)
        y = F.conv2d(xt, torch.from_numpy(w).permute(3, 2, 0, 1).double(), padding=k // 2)
        (gx,) = torch.autograd.grad(y, xt, torch.from_numpy(dy).permute(0, 3, 1, 2).double())
        for path in (4, 5):
            dx = torch.full((B, H, W, ci), float("nan"), device=dev)
            _lib.check(lib.depgan_op_conv2d_bwd_data(P(dyd), P(wd), P(dx), B, H, W, ci, co, k, path, None))
            torch.cuda.synchronize()
            assert rel(dx.cpu().numpy(), gx.permute(0, 2, 3, 1).numpy()) < (TOL if path == 5 else 3e-4)


WGRAD_CASES = [(2, 32, 32, 32, 32, 3), (3, 48, 40, 64, 64, 3), (2, 23, 17, 96, 32, 3), (2, 16, 16, 256, 256, 3),
               (2, 32, 32, 16, 16, 5), (2, 32, 32, 16, 32, 5), (2, 32, 24, 32, 32, 5), (2, 32, 32, 128, 128, 1),
               (2, 32, 32, 1, 32, 3), (2, 32, 32, 2, 32, 3), (2, 32, 32, 1, 16, 5), (4, 64, 64, 32, 64, 3),
               (2, 32, 32, 48, 80, 3)]


@pytest.mark.parametrize("case", WGRAD_CASES)
def test_conv_weight_gradient(lib, case):
    from dep_gan_im_amd import _lib
    B, H, W, ci, co, k = case
    dev = torch.device("cuda:0")
    rng = np.random.default_rng(ci * 77 + co + k)
    x = rng.standard_normal((B, H, W, ci)).astype(np.float32)
    dy = rng.standard_normal((B, H, W, co)).astype(np.float32)
    xd, dyd = torch.from_numpy(x).to(dev), torch.from_numpy(dy).to(dev)
    dw = torch.full((k, k, ci, co), float("nan"), device=dev)
    _lib.check(lib.depgan_op_conv2d_wgrad(P(xd), P(dyd), P(dw), B, H, W, ci, co, k, None))
    torch.cuda.synchronize()
    wt = torch.zeros((co, ci, k, k), dtype=torch.float64, requires_grad=True)
    y = F.conv2d(torch.from_numpy(x).permute(0, 3, 1, 2).double(), wt, padding=k // 2)
    (gw,) = torch.autograd.grad(y, wt, torch.from_numpy(dy).permute(0, 3, 1, 2).double())
    assert rel(dw.cpu().numpy(), gw.permute(2, 3, 1, 0).numpy()) < TOL
    # run-to-run bit reproducibility (no float atomics in the reduction)
    dw2 = torch.empty_like(dw)
    _lib.check(lib.depgan_op_conv2d_wgrad(P(xd), P(dyd), P(dw2), B, H, W, ci, co, k, None))
    torch.cuda.synchronize()
    assert torch.equal(dw, dw2)


WGRAD_BF16_CASES = [(2, 32, 32, 32, 32, 3), (3, 48, 40, 64, 64, 3), (2, 23, 17, 96, 32, 3), (2, 16, 16, 256, 256, 3),
                    (2, 32, 32, 16, 16, 5), (2, 32, 32, 16, 32, 5), (2, 32, 24, 32, 32, 5), (2, 32, 32, 128, 128, 1),
                    (4, 64, 64, 32, 64, 3), (2, 32, 32, 48, 80, 3), (8, 128, 128, 32, 32, 3), (2, 30, 18, 8, 12, 3)]


@pytest.mark.parametrize("case", WGRAD_BF16_CASES)
def test_bf16_mfma_conv_weight_gradient(lib, case):
    """csrc/wgrad_bf16.hip (BASELINE configs[3]: the weight-gradient contraction on v_mfma_f32_32x32x16_bf16, operands
    rounded to bf16 while staged, fp32 accumulation; K-major fragments through ds_read_b64_tr_b16) against the float64
    contraction of the SAME rounded operands: 1e-4, every tap, channel tails, ragged tiles; bit-reproducible."""
    from dep_gan_im_amd import _lib
    B, H, W, ci, co, k = case
    dev = torch.device("cuda:0")
    rng = np.random.default_rng(ci * 79 + co + k)
    x = rng.standard_normal((B, H, W, ci)).astype(np.float32)
    dy = rng.standard_normal((B, H, W, co)).astype(np.float32)
    xd, dyd = torch.from_numpy(x).to(dev), torch.from_numpy(dy).to(dev)
    dw = torch.full((k, k, ci, co), float("nan"), device=dev)
    _lib.check(lib.depgan_op_conv2d_wgrad_bf16(P(xd), P(dyd), P(dw), B, H, W, ci, co, k, None))
    torch.cuda.synchronize()
    wt = torch.zeros((co, ci, k, k), dtype=torch.float64, requires_grad=True)
    y = F.conv2d(torch.from_numpy(_bf16(x)).permute(0, 3, 1, 2).double(), wt, padding=k // 2)
    (gw,) = torch.autograd.grad(y, wt, torch.from_numpy(_bf16(dy)).permute(0, 3, 1, 2).double())
    want = gw.permute(2, 3, 1, 0).numpy()
    assert rel(dw.cpu().numpy(), want) < TOL
    # the rounding is visible (this is not the fp32 contraction) ...
    y32 = F.conv2d(torch.from_numpy(x).permute(0, 3, 1, 2).double(), wt, padding=k // 2)
    (gw32,) = torch.autograd.grad(y32, wt, torch.from_numpy(dy).permute(0, 3, 1, 2).double())
    assert rel(want, gw32.permute(2, 3, 1, 0).numpy()) > 10 * rel(dw.cpu().numpy(), want)
    # ... and the launch is bit-reproducible (deterministic slab reduction)
    dw2 = torch.empty_like(dw)
    _lib.check(lib.depgan_op_conv2d_wgrad_bf16(P(xd), P(dyd), P(dw2), B, H, W, ci, co, k, None))
    torch.cuda.synchronize()
    assert torch.equal(dw, dw2)


DECONV_CASES = [  # B,H,W,Cin,Cout,affine: the three instantiations (64 / 96 / 128 input channels), two channel
    # halves per pixel tile (Cout = 128), image rows shorter than a 32-pixel tile, H != W
    (2, 32, 32, 64, 64, True), (1, 16, 64, 96, 96, True), (2, 16, 16, 128, 128, True), (4, 8, 8, 64, 64, False),
    (1, 4, 32, 64, 128, False), (3, 32, 16, 96, 96, False), (2, 16, 8, 128, 64, True),
]


@pytest.mark.parametrize("case", DECONV_CASES)
def test_fused_transposed_conv_forward(lib, case):
    """deconv_fwd.hip (all four taps of the 2x2 / stride-2 Conv2DTranspose in one workgroup, GT:308) against
    conv_transpose2d in float64, with and without the BN affine + ReLU epilogue."""
    from dep_gan_im_amd import _lib
    B, H, W, ci, co, affine = case
    dev = torch.device("cuda:0")
    rng = np.random.default_rng(ci * 131 + co + H)
    x = rng.standard_normal((B, H, W, ci)).astype(np.float32)
    w = (rng.standard_normal((2, 2, co, ci)) / np.sqrt(ci)).astype(np.float32)      # Keras layout (kh, kw, out, in)
    b = rng.standard_normal(co).astype(np.float32)
    sc = rng.uniform(0.5, 1.5, co).astype(np.float32)
    sh = rng.standard_normal(co).astype(np.float32)
    xd, wd, bd, scd, shd = [torch.from_numpy(a).to(dev) for a in (x, w, b, sc, sh)]
    out = torch.full((B, 2 * H, 2 * W, co), float("nan"), device=dev)
    _lib.check(lib.depgan_op_deconv2x2(P(xd), P(wd), P(bd), P(scd) if affine else None, P(shd) if affine else None,
                                       P(out), B, H, W, ci, co, 1 if affine else 0, None))
    torch.cuda.synchronize()
    # torch: weight (in, out, kh, kw)
    y = F.conv_transpose2d(torch.from_numpy(x).permute(0, 3, 1, 2).double(),
                           torch.from_numpy(w).permute(3, 2, 0, 1).double(), torch.from_numpy(b).double(), stride=2)
    if affine:
        y = torch.relu(y * torch.from_numpy(sc).double()[None, :, None, None] +
                       torch.from_numpy(sh).double()[None, :, None, None])
    assert rel(out.cpu().numpy(), y.permute(0, 2, 3, 1).numpy()) < TOL


@pytest.mark.parametrize("case", [(2, 32, 32, 64, 64), (1, 16, 64, 96, 96), (2, 16, 16, 128, 128), (4, 8, 8, 64, 64),
                                  (32, 16, 16, 96, 96)])
def test_fused_transposed_conv_weight_gradient(lib, case):
    """deconv_wgrad.hip (four taps + the column sums of the upstream gradient in one launch, no LDS) against autograd
    of conv_transpose2d in float64."""
    from dep_gan_im_amd import _lib
    B, H, W, ci, co = case
    dev = torch.device("cuda:0")
    rng = np.random.default_rng(ci * 7 + B + H)
    x = rng.standard_normal((B, H, W, ci)).astype(np.float32)
    dy = rng.standard_normal((B, 2 * H, 2 * W, co)).astype(np.float32)
    xd, dyd = torch.from_numpy(x).to(dev), torch.from_numpy(dy).to(dev)
    dw = torch.full((2, 2, co, ci), float("nan"), device=dev)
    cs = torch.full((co,), float("nan"), device=dev)
    _lib.check(lib.depgan_op_deconv2x2_wgrad(P(xd), P(dyd), P(dw), P(cs), B, H, W, ci, co, None))
    torch.cuda.synchronize()
    wt = torch.zeros(ci, co, 2, 2, dtype=torch.float64, requires_grad=True)
    y = F.conv_transpose2d(torch.from_numpy(x).permute(0, 3, 1, 2).double(), wt, stride=2)
    (gw,) = torch.autograd.grad(y, wt, torch.from_numpy(dy).permute(0, 3, 1, 2).double())
    assert rel(dw.cpu().numpy(), gw.permute(2, 3, 1, 0).numpy()) < TOL          # (in, out, kh, kw) -> (kh, kw, out, in)
    assert rel(cs.cpu().numpy(), dy.astype(np.float64).sum(axis=(0, 1, 2))) < TOL


def test_fused_transposed_conv_refuses_what_it_does_not_cover(lib):
    dev = torch.device("cuda:0")
    x = torch.zeros(1, 8, 8, 48, device=dev)
    w = torch.zeros(2, 2, 48, 48, device=dev)
    out = torch.zeros(1, 16, 16, 48, device=dev)
    assert lib.depgan_op_deconv2x2(P(x), P(w), None, None, None, P(out), 1, 8, 8, 48, 48, 0, None) == 3
    x = torch.zeros(1, 12, 8, 64, device=dev)                      # image height not a power of two
    w = torch.zeros(2, 2, 64, 64, device=dev)
    out = torch.zeros(1, 24, 16, 64, device=dev)
    assert lib.depgan_op_deconv2x2(P(x), P(w), None, None, None, P(out), 1, 12, 8, 64, 64, 0, None) == 3


def test_maxpool(lib):
    from dep_gan_im_amd import _lib
    dev = torch.device("cuda:0")
    x = torch.randn(3, 20, 12, 32, device=dev)
    out = torch.empty(3, 10, 6, 32, device=dev)
    _lib.check(lib.depgan_op_maxpool(P(x), P(out), 3, 10, 6, 32, None))
    torch.cuda.synchronize()
    ref = F.max_pool2d(x.permute(0, 3, 1, 2), 2).permute(0, 2, 3, 1)
    assert torch.equal(out, ref)


@pytest.mark.parametrize("case", [(16, 128, 128, 32, 64, 3), (12, 96, 112, 16, 16, 5), (8, 72, 136, 96, 32, 3),
                                  (16, 128, 128, 32, 64, 3, 6), (8, 72, 136, 96, 32, 3, 6)])
def test_persistent_conv_grid_matches_one_item_per_workgroup(lib, case, monkeypatch):
    """The persistent form of the conv kernel (workgroups looping over items, staging geometry hoisted) must give
    the same bits as the one-item-per-workgroup form, on interior and border tiles (ragged sizes) alike."""
    from dep_gan_im_amd import _lib
    B, H, W, ci, co, k = case[:6]
    path = case[6] if len(case) > 6 else 1          # 6: the 8-channel-chunk form (four resident workgroups per CU)
    dev = torch.device("cuda:0")
    rng = np.random.default_rng(ci + co + k)
    x = torch.from_numpy(rng.standard_normal((B, H, W, ci)).astype(np.float32)).to(dev)
    w = torch.from_numpy((rng.standard_normal((k, k, ci, co)) / np.sqrt(k * k * ci)).astype(np.float32)).to(dev)
    b = torch.from_numpy(rng.standard_normal(co).astype(np.float32)).to(dev)
    outs = []
    for per_cu in ("0", "1", "2"):
        monkeypatch.setenv("DEPGAN_IGEMM_PERSIST", per_cu)      # read by the launcher at every launch
        out = torch.full((B, H, W, co), float("nan"), device=dev)
        _lib.check(lib.depgan_op_conv2d(P(x), P(w), P(b), P(out), B, H, W, ci, co, k, 1, path, None))
        torch.cuda.synchronize()
        outs.append(out.cpu().numpy())
    np.testing.assert_array_equal(outs[0], outs[1])
    np.testing.assert_array_equal(outs[0], outs[2])
    ref = _ref_conv(x[:2].cpu().numpy(), w.cpu().numpy(), b.cpu().numpy(), True)
    assert rel(outs[1][:2], ref) < TOL


@pytest.mark.parametrize("case", [(4, 64, 64, 32, 32), (4, 64, 64, 64, 64), (2, 32, 32, 128, 128), (2, 16, 16, 256, 256)])
def test_winograd_conv_rounding_error_next_to_the_direct_kernel(lib, case):
    """igemm_wino.hip computes the same fp32 contraction with another summation tree (F(2x2,3x3): additions before and
    after 4/9 of the multiplications).  On post-ReLU-like operands of the network's scale: its error against an fp64
    convolution stays within 4 x the direct MFMA kernel's and below 2e-6 of the output's range."""
    from dep_gan_im_amd import _lib
    B, H, W, ci, co = case
    dev = torch.device("cuda:0")
    rng = np.random.default_rng(ci + co)
    x = np.maximum(rng.standard_normal((B, H, W, ci)), 0).astype(np.float32)
    w = (rng.standard_normal((3, 3, ci, co)) * np.sqrt(2.0 / (9 * ci))).astype(np.float32)
    ref = _ref_conv(x, w, None, False)
    xd, wd = torch.from_numpy(x).to(dev), torch.from_numpy(w).to(dev)
    errs = {}
    for path in (1, 8):
        out = torch.full((B, H, W, co), float("nan"), device=dev)
        _lib.check(lib.depgan_op_conv2d(P(xd), P(wd), None, P(out), B, H, W, ci, co, 3, 0, path, None))
        torch.cuda.synchronize()
        e = np.abs(out.cpu().numpy().astype(np.float64) - ref)
        errs[path] = (float(e.max() / np.abs(ref).max()), float(np.sqrt((e ** 2).mean()) / np.abs(ref).max()))
    print("conv %s: direct max %.2e rms %.2e | winograd max %.2e rms %.2e" % (case, *errs[1], *errs[8]))
    assert errs[8][0] < 2e-6 and errs[8][1] < 4.0 * errs[1][1]
