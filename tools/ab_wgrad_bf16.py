"""Timing of the bf16 weight-gradient kernel (wgrad_bf16.hip) against the fp32 one (wgrad.hip) on config 4's shapes.
usage: python tools/ab_wgrad_bf16.py"""
import ctypes as C
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402
from dep_gan_im_amd import _lib  # noqa: E402

lib = _lib.load()
dev = torch.device("cuda:0")


def P(t):
    return C.c_void_p(t.data_ptr())


SHAPES = [(32, 256, 256, 32, 32, 3), (32, 128, 128, 64, 64, 3), (32, 64, 64, 96, 96, 3), (96, 128, 128, 32, 32, 5),
          (96, 256, 256, 16, 16, 5), (96, 128, 128, 16, 32, 5), (96, 64, 64, 64, 64, 3), (96, 16, 16, 256, 256, 3)]
for B, H, W, ci, co, k in SHAPES:
    x = torch.randn(B, H, W, ci, device=dev)
    dy = torch.randn(B, H, W, co, device=dev)
    dw = torch.empty(k, k, ci, co, device=dev)
    res = {}
    for name, fn in (("fp32", lib.depgan_op_conv2d_wgrad), ("bf16", lib.depgan_op_conv2d_wgrad_bf16)):
        for _ in range(2):
            _lib.check(fn(P(x), P(dy), P(dw), B, H, W, ci, co, k, None))
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        N = 5
        for _ in range(N):
            fn(P(x), P(dy), P(dw), B, H, W, ci, co, k, None)
        e1.record()
        torch.cuda.synchronize()
        res[name] = e0.elapsed_time(e1) / N * 1e3
    by = 4.0 * B * H * W * (ci + co)
    print("k%d b%d %dx%d %d->%d: fp32 %.0f us, bf16 %.0f us (%.2f TB/s of operand bytes; includes hipMalloc + slab reduce "
          "of the op entry)" % (k, B, H, W, ci, co, res["fp32"], res["bf16"], by / res["bf16"] / 1e6))
