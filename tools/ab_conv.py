"""Development aid: A/B two code paths of the MFMA conv kernel inside ONE process.  GPU boxes differ by several percent
(clocks), so variants must be compared on the same device, interleaved.  To use it, give ConvArgs a development
switch read from DEPGAN_IGEMM_VAR in dg_conv_igemm (not present in the committed kernels) and branch on it."""
import os, sys, ctypes as C, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from dep_gan_im_amd import _lib
lib = _lib.load(); dev = torch.device("cuda:0")
def P(t): return C.c_void_p(t.data_ptr())
shapes = [(32,256,256,32,32,3), (32,128,128,64,64,3), (32,256,256,96,32,3), (96,256,256,16,16,5)]
if os.environ.get("AB_SHAPES"):     # e.g. AB_SHAPES="32,128,128,64,64,3;96,16,16,256,256,3"
    shapes = [tuple(int(v) for v in t.split(",")) for t in os.environ["AB_SHAPES"].split(";")]
variants = [int(v) for v in (sys.argv[1:] or ["0", "1"])]
for (B,H,W,ci,co,k) in shapes:
    x = torch.randn(B,H,W,ci, device=dev); w = torch.randn(k,k,ci,co, device=dev)*0.05; out = torch.empty(B,H,W,co, device=dev)
    res = {v: [] for v in variants}
    for rnd in range(6):
        for v in variants:
            os.environ[os.environ.get("AB_KEY", "DEPGAN_IGEMM_VAR")] = str(v)
            lib.depgan_op_conv2d_stamps(P(x),P(w),P(out),B,H,W,ci,co,k,None,3,None); torch.cuda.synchronize()
            t = time.perf_counter(); lib.depgan_op_conv2d_stamps(P(x),P(w),P(out),B,H,W,ci,co,k,None,30,None); torch.cuda.synchronize()
            res[v].append((time.perf_counter() - t) / 30 * 1e6)
    print("k%d b%d %dx%d %d->%d: " % (k,B,H,W,ci,co) + "  ".join("var%d %.1f us (min %.1f)" % (v, sum(res[v][1:]) / (len(res[v]) - 1), min(res[v])) for v in variants))
