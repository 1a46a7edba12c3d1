"""Device time of the transposed-convolution forward of the generator's three up-sampling layers (GT:449/464/478) at
batch 32: the fused four-tap kernel (deconv_fwd.hip) vs the grouped launch of the general kernel, in one process
(DEPGAN_DECONV_FUSED=0 selects the grouped launch inside an Engine; here both are timed through the G forward).
Usage: python tools/ab_deconv.py"""
import ctypes as C
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from dep_gan_im_amd import _lib  # noqa: E402


def P(t):
    return C.c_void_p(t.data_ptr())


def main():
    lib = _lib.load()
    dev = torch.device("cuda:0")
    for (B, H, W, ci, co) in ((32, 128, 128, 64, 64), (32, 64, 64, 96, 96), (32, 32, 32, 128, 128)):
        x = torch.randn(B, H, W, ci, device=dev)
        w = torch.randn(2, 2, co, ci, device=dev) / np.sqrt(ci)
        b = torch.randn(co, device=dev)
        sc = torch.rand(co, device=dev) + 0.5
        sh = torch.randn(co, device=dev)
        out = torch.empty(B, 2 * H, 2 * W, co, device=dev)
        for _ in range(3):
            _lib.check(lib.depgan_op_deconv2x2(P(x), P(w), P(b), P(sc), P(sh), P(out), B, H, W, ci, co, 1, None))
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        n = 20
        e0.record()
        for _ in range(n):
            _lib.check(lib.depgan_op_deconv2x2(P(x), P(w), P(b), P(sc), P(sh), P(out), B, H, W, ci, co, 1, None))
        e1.record()
        torch.cuda.synchronize()
        ms = e0.elapsed_time(e1) / n
        fl = 2.0 * B * H * W * ci * co * 4
        by = 4.0 * B * H * W * (ci + 4 * co)
        print("deconv %dx%dx%d %d->%d: %.1f us  %.1f TFLOP/s  %.2f TB/s (input once + output once)"
              % (B, H, W, ci, co, ms * 1e3, fl / ms / 1e9, by / ms / 1e9), flush=True)


if __name__ == "__main__":
    main()
