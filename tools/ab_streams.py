"""Development aid: do two independent MFMA conv launches on two streams finish sooner than back to back on one?"""
import os, sys, ctypes as C, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from dep_gan_im_amd import _lib
lib = _lib.load(); dev = torch.device("cuda:0")
def P(t): return C.c_void_p(t.data_ptr())
def mk(B,H,W,ci,co,k):
    return (torch.randn(B,H,W,ci, device=dev), torch.randn(k,k,ci,co, device=dev)*0.05, torch.empty(B,H,W,co, device=dev), (B,H,W,ci,co,k))
pairs = [((32,256,256,32,32,3),(32,256,256,32,32,3)), ((32,128,128,64,64,3),(32,256,256,32,32,3)), ((32,16,16,256,256,3),(32,32,32,128,128,3)),
         ((32,16,16,256,256,3),(32,16,16,256,256,3))]
s1, s2 = torch.cuda.Stream(), torch.cuda.Stream()
for sa, sb in pairs:
    A, Bb = mk(*sa), mk(*sb)
    def run(t, stream, reps):
        x,w,o,(B,H,W,ci,co,k) = t
        lib.depgan_op_conv2d_stamps(P(x),P(w),P(o),B,H,W,ci,co,k,None,reps,C.c_void_p(stream.cuda_stream))
    # note: depgan_op_conv2d_stamps synchronises its stream at the end (packs weights, frees) -> use threads for overlap
    import threading
    res = {}
    for mode in ("serial", "concurrent"):
        ts = []
        for rnd in range(4):
            torch.cuda.synchronize(); t0 = time.perf_counter()
            if mode == "serial":
                run(A, s1, 20); run(Bb, s1, 20)
            else:
                th = [threading.Thread(target=run, args=(A, s1, 20)), threading.Thread(target=run, args=(Bb, s2, 20))]
                [t.start() for t in th]; [t.join() for t in th]
            torch.cuda.synchronize(); ts.append((time.perf_counter() - t0) / 20 * 1e6)
        res[mode] = min(ts)
    print(sa, "+", sb, ": serial %.1f us per pair, concurrent %.1f us (x%.3f)" % (res["serial"], res["concurrent"], res["serial"]/res["concurrent"]))
