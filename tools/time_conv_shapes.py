"""Kernel-only time of the step's large convolution shapes through the operator entry of whichever library DEPGAN_LIB
names (A/B of two builds on one box: run it alternately under both).  usage: DEPGAN_LIB=<so> python tools/time_conv_shapes.py"""
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


SHAPES = [(32, 256, 256, 32, 32, 3, 6), (32, 128, 128, 64, 64, 3, 6), (32, 64, 64, 96, 96, 3, 6), (32, 256, 256, 96, 32, 3, 6),
          (96, 256, 256, 16, 16, 5, 1), (96, 128, 128, 32, 32, 5, 1), (96, 16, 16, 256, 256, 3, 1)]
out_line = []
for B, H, W, ci, co, k, path in SHAPES:
    x = torch.randn(B, H, W, ci, device=dev)
    w = torch.randn(k, k, ci, co, device=dev) * 0.05
    b = torch.zeros(co, device=dev)
    out = torch.empty(B, H, W, co, device=dev)
    best = 1e9
    for rep in range(3):
        for _ in range(2):
            _lib.check(lib.depgan_op_conv2d(P(x), P(w), P(b), P(out), B, H, W, ci, co, k, 1, path, None))
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        N = 10
        for _ in range(N):
            lib.depgan_op_conv2d(P(x), P(w), P(b), P(out), B, H, W, ci, co, k, 1, path, None)
        e1.record()
        torch.cuda.synchronize()
        best = min(best, e0.elapsed_time(e1) / N * 1e3)
    fl = 2.0 * B * H * W * ci * co * k * k
    out_line.append("k%d b%d %dx%d %d->%d %.1f us (%.3f)" % (k, B, H, W, ci, co, best, fl / best / 1e6 / 157.3))
print(os.environ.get("DEPGAN_LIB", "default"), " | ".join(out_line))
