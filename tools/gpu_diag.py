"""GPU diagnostic sweep (development aid): runs every operator / closure check,
prints max errors vs the CPU oracle and keeps going after a failure."""
import os, sys, time, traceback, ctypes as C
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import torch.nn.functional as F
from dep_gan_im_amd import _lib, Engine
from oracle import depgan_oracle as O

lib = _lib.load()
dev = torch.device("cuda:0")
RES = []

def rel(a, b):
    a = np.asarray(a, np.float64); b = np.asarray(b, np.float64)
    return float(np.abs(a - b).max() / (np.abs(b).max() + 1e-30))

def report(name, err, tol=1e-4):
    ok = err < tol
    RES.append((name, err, ok))
    print("%-60s %.3e %s" % (name, err, "ok" if ok else "FAIL"), flush=True)

def P(t): return C.c_void_p(t.data_ptr())

def conv_ref(x, w, b, relu):
    y = F.conv2d(torch.from_numpy(x).permute(0,3,1,2).double(), torch.from_numpy(w).permute(3,2,0,1).double(),
                 None if b is None else torch.from_numpy(b).double(), padding=w.shape[0]//2)
    if relu: y = torch.relu(y)
    return y.permute(0,2,3,1).numpy()

def test_conv(B,H,W,ci,co,k,path,relu=1):
    rng = np.random.default_rng(ci*1000+co+k)
    x = rng.standard_normal((B,H,W,ci)).astype(np.float32); w = (rng.standard_normal((k,k,ci,co))/np.sqrt(k*k*ci)).astype(np.float32)
    b = rng.standard_normal(co).astype(np.float32)
    xd, wd, bd = [torch.from_numpy(a).to(dev) for a in (x,w,b)]
    out = torch.full((B,H,W,co), float('nan'), device=dev)
    _lib.check(lib.depgan_op_conv2d(P(xd),P(wd),P(bd),P(out),B,H,W,ci,co,k,relu,path,None), "conv")
    torch.cuda.synchronize()
    report("conv fwd B%d %dx%d %d->%d k%d path%d" % (B,H,W,ci,co,k,path), rel(out.cpu().numpy(), conv_ref(x,w,b,relu)))
    # bwd data
    dy = rng.standard_normal((B,H,W,co)).astype(np.float32)
    dyd = torch.from_numpy(dy).to(dev); dx = torch.full((B,H,W,ci), float('nan'), device=dev)
    try:
        _lib.check(lib.depgan_op_conv2d_bwd_data(P(dyd),P(wd),P(dx),B,H,W,ci,co,k,path,None), "bwd")
        torch.cuda.synchronize()
        xt = torch.from_numpy(x).permute(0,3,1,2).double().requires_grad_(True)
        y = F.conv2d(xt, torch.from_numpy(w).permute(3,2,0,1).double(), padding=k//2)
        (gx,) = torch.autograd.grad(y, xt, torch.from_numpy(dy).permute(0,3,1,2).double())
        report("conv bwd-data  %d->%d k%d path%d" % (ci,co,k,path), rel(dx.cpu().numpy(), gx.permute(0,2,3,1).numpy()))
    except Exception as e:
        print("   bwd-data skipped/failed:", e)

def test_wgrad(B,H,W,ci,co,k):
    rng = np.random.default_rng(ci*77+co+k)
    x = rng.standard_normal((B,H,W,ci)).astype(np.float32); dy = rng.standard_normal((B,H,W,co)).astype(np.float32)
    xd, dyd = torch.from_numpy(x).to(dev), torch.from_numpy(dy).to(dev)
    dw = torch.full((k,k,ci,co), float('nan'), device=dev)
    _lib.check(lib.depgan_op_conv2d_wgrad(P(xd),P(dyd),P(dw),B,H,W,ci,co,k,None), "wgrad")
    torch.cuda.synchronize()
    wt = torch.zeros((co,ci,k,k), dtype=torch.float64, requires_grad=True)
    y = F.conv2d(torch.from_numpy(x).permute(0,3,1,2).double(), wt, padding=k//2)
    (gw,) = torch.autograd.grad(y, wt, torch.from_numpy(dy).permute(0,3,1,2).double())
    report("wgrad B%d %dx%d %d->%d k%d" % (B,H,W,ci,co,k), rel(dw.cpu().numpy(), gw.permute(2,3,1,0).numpy()))

def section(fn, *a):
    try: fn(*a)
    except Exception as e:
        traceback.print_exc(); RES.append((fn.__name__+str(a), float('nan'), False))

if "ops" in sys.argv or len(sys.argv) == 1:
    for args in [(2,32,32,32,32,3,1),(2,48,40,32,64,3,1),(1,32,32,96,96,3,1),(1,32,32,224,96,3,1),(2,32,32,64,160,3,1),
                 (2,32,32,16,16,5,1),(2,32,32,16,32,5,1),(2,32,32,32,32,5,1),(2,32,32,32,16,5,1),(2,32,32,128,128,1,1),(2,32,32,64,16,1,1),
                 (2,32,32,48,48,3,1),
                 (2,32,32,1,32,3,2),(2,32,32,2,32,3,2),(2,32,32,1,16,5,2),(2,32,32,16,1,5,2),(2,32,32,8,8,3,2),(1,16,16,256,256,3,1)]:
        section(test_conv, *args)
    for args in [(2,32,32,32,32,3),(3,48,40,64,64,3),(2,32,32,96,32,3),(2,16,16,256,256,3),(2,32,32,16,16,5),(2,32,32,16,32,5),
                 (2,32,32,32,32,5),(2,32,32,128,128,1),(2,32,32,1,32,3),(2,32,32,2,32,3),(2,32,32,1,16,5),(4,64,64,32,64,3),(2,32,32,48,80,3)]:
        section(test_wgrad, *args)

def model_checks(img, B, seed=1):
    PG = O.init_generator(seed, bias_std=0.05); PD = O.init_critic(seed+1, bias_std=0.05, img=img); PD2 = O.init_critic(seed+2, bias_std=0.05, img=img)
    x, y2, z, ep = O.synth_batch(seed+5, B, img, img)
    eng = Engine(B, img, img, 1)
    eng.set_weights("G", PG); eng.set_weights("D_y2", PD); eng.set_weights("D_dem", PD2)
    w = eng.get_weights("G")
    report("[%d] weights roundtrip" % img, max(rel(w[k], PG[k]) for k in PG), 1e-7)
    a = eng.g_forward(x, z).cpu().numpy()
    a_ref = O.g_predict(PG, x, z)
    report("[%d] G forward" % img, rel(a, a_ref), 1e-3)
    d = eng.d_forward("D_y2", y2).cpu().numpy(); d_ref = O.d_predict(PD, y2)
    report("[%d] D forward" % img, rel(d, d_ref), 1e-3)
    # generator eval
    g = eng.generator(x, y2, z, "eval"); g_ref = O.g_eval(PG, PD, PD2, x, y2, z)
    print("   g_eval", g, "\n   ref   ", g_ref)
    report("[%d] netG_no_update" % img, max(abs(a_-b_)/(abs(b_)+1e-3) for a_, b_ in zip(g, g_ref)), 1e-3)
    # critic grads
    for which, PDx in (("D_y2", PD), ("D_dem", PD2)):
        out = eng.critic(which, y2, x, z, ep, update=False)
        outs, grads, aux = O.critic_grads(PDx, PG, y2, x, z, ep, "y2" if which == "D_y2" else "dem")
        print("   critic", which, out, outs, "gp sums", eng.last_sums()[:4], aux["gp"])
        report("[%d] %s outs" % (img, which), max(abs(a_-b_)/(abs(b_)+1e-3) for a_, b_ in zip(out, outs)), 1e-3)
        gg = eng.get_grads(which)
        worst = 0
        for k in grads:
            e = rel(gg[k], grads[k]); worst = max(worst, e)
            if e > 1e-3: print("      grad", k, e, np.abs(grads[k]).max())
        report("[%d] %s grads (worst tensor)" % (img, which), worst, 1e-3)
    # generator grads
    g = eng.generator(x, y2, z, "grads"); outs, grads = O.g_grads(PG, PD, PD2, x, y2, z)
    report("[%d] netG_train outs" % img, max(abs(a_-b_)/(abs(b_)+1e-3) for a_, b_ in zip(g, outs)), 1e-3)
    gg = eng.get_grads("G"); worst = 0
    for k in grads:
        e = rel(gg[k], grads[k]); worst = max(worst, e)
        if e > 1e-3: print("      grad", k, e, np.abs(grads[k]).max())
    report("[%d] G grads (worst tensor)" % img, worst, 1e-3)
    # full steps with weight comparison
    tr = O.OracleTrainers(PG, PD, PD2)
    o1 = eng.critic("D_y2", y2, x, z, ep); r1 = tr.netD_y2_train([y2, x, z, ep])
    o2 = eng.critic("D_dem", y2, x, z, ep); r2 = tr.netD_dem_train([y2, x, z, ep])
    o3 = eng.generator(x, y2, z, "step"); r3 = tr.netG_train([x, y2, z])
    for net, Pn in (("D_y2", PD), ("D_dem", PD2), ("G", PG)):
        w = eng.get_weights(net)
        report("[%d] post-step weights %s" % (img, net), max(np.abs(w[k]-Pn[k]).max() for k in Pn), 1e-3)
    o4 = eng.generator(x, y2, z, "eval"); r4 = tr.netG_no_update([x, y2, z])
    report("[%d] eval after steps" % img, max(abs(a_-b_)/(abs(b_)+1e-3) for a_, b_ in zip(o4, r4)), 1e-3)
    eng.close()

if "model" in sys.argv or len(sys.argv) == 1:
    section(model_checks, 64, 2)
    section(model_checks, 256, 2, 3)

if "perf" in sys.argv:
    B = int(os.environ.get("DG_B", "32"))
    eng = Engine(B, 256, 256, 1)
    PG = O.init_generator(1); PD = O.init_critic(2); PD2 = O.init_critic(3)
    eng.set_weights("G", PG); eng.set_weights("D_y2", PD); eng.set_weights("D_dem", PD2)
    x, y2, z, ep = O.synth_batch(7, B)
    xd, y2d, zd, epd = [torch.from_numpy(a).to(dev) for a in (x, y2, z, ep)]
    for it in range(2):
        eng.critic("D_y2", y2d, xd, zd, epd); eng.critic("D_dem", y2d, xd, zd, epd); eng.generator(xd, y2d, zd, "step")
    torch.cuda.synchronize()
    def timeit(fn, n=3):
        torch.cuda.synchronize(); t = time.time()
        for _ in range(n): fn()
        torch.cuda.synchronize(); return (time.time() - t) / n * 1e3
    tf = timeit(lambda: eng.g_forward(xd, zd))
    print("G forward B=%d: %.2f ms -> %.1f TF/s (%.1f%% of 157.3)" % (B, tf, 23.513e9*B/tf/1e9, 23.513e9*B/tf/1e9/157.3*100))
    tc = timeit(lambda: eng.critic("D_y2", y2d, xd, zd, epd))
    print("critic step: %.2f ms -> %.1f TF/s" % (tc, 61.62e9*B/tc/1e9))
    te = timeit(lambda: eng.generator(xd, y2d, zd, "eval"))
    print("G eval: %.2f ms -> %.1f TF/s" % (te, 31.14e9*B/te/1e9))
    tg = timeit(lambda: eng.generator(xd, y2d, zd, "step"))
    print("G step: %.2f ms -> %.1f TF/s" % (tg, 85.78e9*B/tg/1e9))
    tot = 2*tc + tg
    print("canonical step: %.2f ms -> %.1f slices/s, %.1f TF/s" % (tot, B/tot*1e3, 209.0e9*B/tot/1e9))
    eng.profile(True); eng.profile_reset()
    eng.critic("D_y2", y2d, xd, zd, epd); eng.critic("D_dem", y2d, xd, zd, epd); eng.generator(xd, y2d, zd, "step")
    for k, nm in ((0, "mfma conv"), (1, "mfma wgrad"), (2, "other")):
        ms, n, fl = eng.profile_read(k)
        print("class %-10s %8.2f ms %5d launches %8.1f TF/s" % (nm, ms, n, fl/ms/1e9 if ms else 0))
    eng.profile_dump("gpurun_out/launches.csv")

nfail = sum(1 for r in RES if not r[2])
print("\n%d checks, %d failed" % (len(RES), nfail))
