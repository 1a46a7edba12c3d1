"""A/B of the wave-private 3x3 kernel (path 7) against the workgroup-tile kernel (path 6) on the step's shapes, same
process, interleaved.  usage: python tools/ab_wp.py"""
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


SHAPES = [(32, 256, 256, 32, 32), (32, 128, 128, 64, 64), (32, 128, 128, 32, 64), (96, 64, 64, 64, 64), (96, 64, 64, 32, 64)]
for B, H, W, ci, co in SHAPES:
    x = torch.randn(B, H, W, ci, device=dev)
    w = torch.randn(3, 3, ci, co, device=dev) * 0.05
    b = torch.zeros(co, device=dev)
    out = torch.empty(B, H, W, co, device=dev)
    res = {}
    for rep in range(3):
        for path in (6, 7):
            # (the op entry packs the weights on every call: a few microseconds on these shapes, same for both paths)
            for _ in range(2):
                _lib.check(lib.depgan_op_conv2d(P(x), P(w), P(b), P(out), B, H, W, ci, co, 3, 1, path, None))
            torch.cuda.synchronize()
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            N = 10
            for _ in range(N):
                lib.depgan_op_conv2d(P(x), P(w), P(b), P(out), B, H, W, ci, co, 3, 1, path, None)
            e1.record()
            torch.cuda.synchronize()
            res.setdefault(path, []).append(e0.elapsed_time(e1) / N * 1e3)
    fl = 2.0 * B * H * W * ci * co * 9
    t6, t7 = min(res[6]), min(res[7])
    print("b%d %dx%d %d->%d: tile %.1f us (%.1f TF, %.3f)  wave-private %.1f us (%.1f TF, %.3f)  %+.1f %%"
          % (B, H, W, ci, co, t6, fl / t6 / 1e6, fl / t6 / 1e6 / 157.3, t7, fl / t7 / 1e6, fl / t7 / 1e6 / 157.3,
             100 * (t7 - t6) / t6))
