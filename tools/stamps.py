"""Development aid: per-workgroup phase timeline of the MFMA conv kernel."""
import os, sys, ctypes as C
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from dep_gan_im_amd import _lib
lib = _lib.load(); dev = torch.device("cuda:0")
def P(t): return C.c_void_p(t.data_ptr())
def run(B,H,W,ci,co,k):
    x = torch.randn(B,H,W,ci, device=dev); w = torch.randn(k,k,ci,co, device=dev)*0.05; out = torch.empty(B,H,W,co, device=dev)
    nwg = ((H+15)//16)*((W+15)//16)*B*((co+31)//32)
    st = torch.zeros(nwg*16, dtype=torch.int64, device=dev)
    _lib.check(lib.depgan_op_conv2d_stamps(P(x),P(w),P(out),B,H,W,ci,co,k,P(st),3,None))
    torch.cuda.synchronize()
    s = st.cpu().numpy().reshape(nwg,16).astype(np.int64)
    t0 = s[:,0].min()
    dur = s[:,5]-s[:,0]
    clk = dur/np.maximum(s[:,6],1)*100.0  # MHz (realtime counter is 100 MHz)
    print("== B%d %dx%d %d->%d k%d: %d WGs; kernel span %.1f us (cycles %d); clock ~%.0f MHz" % (B,H,W,ci,co,k,nwg,(s[:,5].max()-t0)/np.median(clk), s[:,5].max()-t0, np.median(clk)))
    names = ["prefetch-issue","stage0-ready","stage1-ready","mfma-done","epilogue"]
    prev = s[:,0]
    for i,nm in enumerate(names):
        cur = s[:,i+1]
        ok = cur>0
        d = (cur-prev)[ok]
        print("   %-16s median %7d  p10 %7d  p90 %7d cycles" % (nm, np.median(d), np.percentile(d,10), np.percentile(d,90)))
        prev = np.where(ok, cur, prev)
    print("   WG lifetime     median %7d p10 %7d p90 %7d" % (np.median(dur), np.percentile(dur,10), np.percentile(dur,90)))
    # concurrency per CU: HW_ID bits: cu_id [11:8], sh_id [12], se_id [15:13]?; just count distinct ids vs WGs alive at the median time
    hw = s[:,7]
    tmid = t0 + (s[:,5].max()-t0)//2
    alive = (s[:,0] <= tmid) & (s[:,5] >= tmid)
    print("   alive at mid-kernel: %d WGs (%.2f per CU if 256 CUs)" % (alive.sum(), alive.sum()/256.0))
    starts = np.sort(s[:,0]-t0)
    print("   start times: first-wave (768th WG) at %d cycles; median start %d" % (starts[min(767,len(starts)-1)], np.median(starts)))
import time
def timeit(B,H,W,ci,co,k,n=20):
    x = torch.randn(B,H,W,ci, device=dev); w = torch.randn(k,k,ci,co, device=dev)*0.05; out = torch.empty(B,H,W,co, device=dev)
    lib.depgan_op_conv2d_stamps(P(x),P(w),P(out),B,H,W,ci,co,k,None,3,None); torch.cuda.synchronize()
    t=time.time(); lib.depgan_op_conv2d_stamps(P(x),P(w),P(out),B,H,W,ci,co,k,None,n,None); torch.cuda.synchronize(); dt=(time.time()-t)/n
    print("   time %.1f us -> %.1f TF/s" % (dt*1e6, 2.0*B*H*W*ci*co*k*k/dt/1e12))
cases = [(32,256,256,32,32,3),(32,128,128,64,64,3),(32,64,64,96,96,3),(32,256,256,96,32,3)]
if os.environ.get('ONEWG'): cases = [(1,256,256,32,32,3)]
for args in cases:
    if os.environ.get("STAMPS","1") == "1": run(*args)
    timeit(*args)
