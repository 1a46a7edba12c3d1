"""Development aid: run one MFMA conv shape a few times (for rocprofv3 --pmc)."""
import os, sys, ctypes as C
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from dep_gan_im_amd import _lib
lib = _lib.load(); dev = torch.device("cuda:0")
def P(t): return C.c_void_p(t.data_ptr())
B,H,W,ci,co,k = [int(v) for v in (sys.argv[1:7] if len(sys.argv) > 6 else (32,256,256,32,32,3))]
x = torch.randn(B,H,W,ci, device=dev); w = torch.randn(k,k,ci,co, device=dev)*0.05; out = torch.empty(B,H,W,co, device=dev)
lib.depgan_op_conv2d_stamps(P(x),P(w),P(out),B,H,W,ci,co,k,None,5,None); torch.cuda.synchronize()
