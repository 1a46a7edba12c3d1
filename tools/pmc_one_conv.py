"""Development aid: run one MFMA conv shape a few times through the library's default path (for rocprofv3 --pmc)."""
import os, sys, ctypes as C
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from dep_gan_im_amd import _lib
lib = _lib.load(); dev = torch.device("cuda:0")
def P(t): return C.c_void_p(t.data_ptr())
B,H,W,ci,co,k = [int(v) for v in (sys.argv[1:7] if len(sys.argv) > 6 else (32,256,256,32,32,3))]
x = torch.randn(B,H,W,ci, device=dev); w = torch.randn(k,k,ci,co, device=dev)*0.05; out = torch.empty(B,H,W,co, device=dev)
b = torch.zeros(co, device=dev)
for _ in range(3):
    _lib.check(lib.depgan_op_conv2d(P(x),P(w),P(b),P(out),B,H,W,ci,co,k,int(os.environ.get("RELU","1")),int(os.environ.get("CONV_PATH","1")),None))
torch.cuda.synchronize()

import time
N = 20
t0 = time.perf_counter()
for _ in range(N):
    lib.depgan_op_conv2d(P(x),P(w),P(b),P(out),B,H,W,ci,co,k,int(os.environ.get("RELU","1")),int(os.environ.get("CONV_PATH","1")),None)
torch.cuda.synchronize()
us = (time.perf_counter() - t0) / N * 1e6
print("conv k%d b%d %dx%d %d->%d: %.1f us  %.1f TF/s (pack + conv per call)" % (k, B, H, W, ci, co, us, 2.0*B*H*W*ci*co*k*k/us/1e6))
