import os, sys, ctypes as C
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
import numpy as np, torch
from dep_gan_im_amd import _lib
lib = _lib.load(); dev = torch.device("cuda:0")
def P(t): return C.c_void_p(t.data_ptr())
for (B,H,W,ci,co,k) in [(1,256,256,32,32,3),(32,256,256,32,32,3)]:
    x = torch.randn(B,H,W,ci, device=dev); w = torch.randn(k,k,ci,co, device=dev)*0.05; out = torch.empty(B,H,W,co, device=dev)
    nwg = (H//16)*(W//16)*B*(co//32)
    st = torch.zeros(nwg*16, dtype=torch.int64, device=dev)
    _lib.check(lib.depgan_op_conv2d_stamps(P(x),P(w),P(out),B,H,W,ci,co,k,P(st),3,None)); torch.cuda.synchronize()
    s = st.cpu().numpy().reshape(nwg,16)
    for nm,a,b in (("barrier",4,13),("scatter",13,14),("read+store",14,5)):
        d = s[:,b]-s[:,a]; print(B, nm, int(np.median(d)), int(np.percentile(d,10)), int(np.percentile(d,90)))
