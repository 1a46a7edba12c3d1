#!/usr/bin/env python3
"""Headline benchmark: 2-D slices/sec of the DEP-GAN two-critic WGAN-GP train step.

One "step" = the canonical unit of SURVEY.md 8(d): one critic-Y2 update, one
critic-DEM update and one generator update (GT:809, 824, 878) on one batch of
synthetic 256x256x1 slices, fp32, per-GPU batch 32 (BASELINE.json configs[1]).
Inputs are resident in HBM before the timed region.  --gpus N > 1: one rank per
GPU (weak scaling, per-GPU batch fixed), each network update all-reduces its
flat gradient arena (+ loss pieces) over RCCL.  Either the driver launches this
file under torch.distributed.run (WORLD_SIZE set), or -- `python bench.py
--gpus N` on its own -- this process starts that launcher as a CHILD before
anything touches HIP, relays rank 0's JSON line and exits with the child's
status.  --dry-run replaces the GPU engine by a host stand-in and RCCL by gloo
so that the launcher / rank plumbing can be rehearsed on a machine without GPUs.

Extra objects on the same line at N = 1 (each builds an engine of its own after the timed region, none is the headline):
  config4:   BASELINE configs[3] on the bf16 matrix pipe (256x256x2, bf16 weights and activations, fp32 accumulate)
  config5:   BASELINE configs[4], the DEP-UResNet supervised step
  f32_split: the headline workload with the opt-in split-product convolutions (fp32 operands as exact sums of bf16
             terms, six / three cross products on the bf16 pipe)
  direct_conv: the headline workload with DEPGAN_WINOGRAD=0 (3x3 layers on the direct implicit-GEMM kernel)

Prints ONE JSON line on rank 0 (contract in the task statement), including
  roofline:     MFMA implicit-GEMM convolution class (dominant kernels) --
                algorithmic FLOPs / HIP-event time, measured live on extra steps
                outside the timed region, against the 157.3 TFLOP/s fp32 matrix peak
  cpu_baseline: the CPU oracle (a port: the Keras/TF reference cannot run) timed
                on this box's host cores on a bounded sample of the same workload.
"""
import argparse
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)
# the host driver only supports dmabuf IPC: must be in the environment before the first HIP call of the process
os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")

GFLOP_PER_SLICE = 209.0          # canonical step, SURVEY 8(d)
PEAK_F32_MFMA = 157.3            # TFLOP/s, MI355X_MICROARCH.md chip table


def synth(seed, B, H=256, W=256):
    """Same construction as SURVEY App. C (kept here so the product path does not import oracle/)."""
    rng = np.random.default_rng(seed)
    yy, xx = np.mgrid[0:H, 0:W].astype(np.float64)
    mask = (((yy - 128) / 100.0) ** 2 + ((xx - 128) / 80.0) ** 2) <= 1.0
    x0 = np.zeros((B, H, W))
    y2 = np.zeros((B, H, W))
    for b in range(B):
        K = int(rng.integers(3, 9))
        cy, cx = 128 + rng.uniform(-70, 70, K), 128 + rng.uniform(-55, 55, K)
        amp, sig = rng.uniform(0.3, 1.0, K), rng.uniform(2.0, 8.0, K)
        amp2, sig2 = amp * rng.uniform(0.7, 1.3, K), sig * rng.uniform(0.7, 1.3, K)
        for k in range(K):
            d2 = (yy - cy[k]) ** 2 + (xx - cx[k]) ** 2
            x0[b] += amp[k] * np.exp(-d2 / (2 * sig[k] ** 2))
            y2[b] += amp2[k] * np.exp(-d2 / (2 * sig2[k] ** 2))
        x0[b] = np.clip(x0[b] + 0.15 * rng.uniform(size=(H, W)) ** 4, 0, 1) * mask
        y2[b] = np.clip(y2[b] + 0.15 * rng.uniform(size=(H, W)) ** 4, 0, 1) * mask
    z = rng.normal(size=(B, 32, 1))
    ep = rng.uniform(size=(B, 1, 1, 1))
    return [a.astype(np.float32) for a in (x0[..., None], y2[..., None], z, ep)]


def cpu_baseline(sample_batch):
    """Oracle (port) timed on the host cores: one canonical step on `sample_batch` slices."""
    import torch
    from oracle import depgan_oracle as O
    try:
        cores = len(os.sched_getaffinity(0))
    except AttributeError:
        cores = os.cpu_count() or 1
    cores = max(1, min(cores, 16))   # the GPU box gives one GPU's share of the host (16 cores)
    torch.set_num_threads(cores)
    PG, PD1, PD2 = O.init_generator(1), O.init_critic(2), O.init_critic(3)
    x, y2, z, ep = O.synth_batch(7, sample_batch)
    tr = O.OracleTrainers(PG, PD1, PD2)
    t0 = time.time()
    tr.netD_y2_train([y2, x, z, ep])
    tr.netD_dem_train([y2, x, z, ep])
    tr.netG_train([x, y2, z])
    dt = time.time() - t0
    # BASELINE.json configs[0] / BASELINE.md section 3 step 2: generator forward only, batch 4, median of 5
    x4, _, z4, _ = O.synth_batch(8, 4)
    O.g_predict(PG, x4, z4)
    ts = []
    for _ in range(5):
        t1 = time.time()
        O.g_predict(PG, x4, z4)
        ts.append(time.time() - t1)
    med = sorted(ts)[2]
    return {"value": round(sample_batch / dt, 4), "unit": "slices/s", "cores": cores, "kind": "port",
            "sample": "one canonical step (critic-Y2 + critic-DEM + G update) at batch %d, 256x256x1 fp32, "
                      "torch-CPU restatement of the Keras graph (oracle/depgan_oracle.py), %.1f s" % (sample_batch, dt),
            "g_forward_b4": {"value": round(4 / med, 3), "unit": "slices/s", "ms": round(med * 1e3, 1),
                             "sample": "BASELINE configs[0]: netG forward only, batch 4, 256x256x1, median of 5"}}


def pmc_traffic(batch, launches_per_step):
    """HBM bytes per launch of the convolution class from the committed PMC passes (tools/pmc_traffic.py; counters
    cannot be read from inside the process being timed).  The passes run this file with DEPGAN_BENCH_STEP_ONLY=1, i.e.
    over canonical steps and nothing else, so the population is the one `algorithmic_bytes_per_launch` describes: the
    file is refused unless its launch count is a whole number of steps of THIS build (a kernel added to or removed from
    the step since the passes were taken makes the count stop dividing)."""
    best = None
    for f in sorted(os.listdir(os.path.join(ROOT, "profiles"))):
        if f.endswith("pmc_traffic.json"):
            best = f
    if best is None or batch != 32:
        return None, "no PMC pass for this configuration"
    with open(os.path.join(ROOT, "profiles", best)) as fh:
        d = json.load(fh)
    cls = d["igemm_conv_kernel_class"]
    n = int(cls.get("launches", 0))
    if not d.get("step_only") or launches_per_step <= 0 or n == 0 or n % launches_per_step:
        return None, ("profiles/%s refused: %d class launches is not a whole number of canonical steps of %d launches "
                      "(or the passes were not taken with DEPGAN_BENCH_STEP_ONLY=1)" % (best, n, launches_per_step))
    return round(cls["hbm_bytes_per_launch"]), "profiles/%s, %d launches = %d steps x %d" % (
        best, n, n // launches_per_step, launches_per_step)


def issued_share(kernel_name):
    """MFMA flops a kernel issues per algorithmic flop of its convolution: 4/9 for the Winograd F(2x2,3x3) kernels
    (csrc/igemm_wino.hip: 16 multiplications per 2x2 outputs and input channel instead of 36), 1 for the direct ones."""
    return 4.0 / 9.0 if kernel_name.startswith("wino_") else 1.0


def dominant_kernel(eng, steps):
    """The convolution class split by kernel instantiation (the names rocprofv3 --kernel-trace --stats prints), from the
    HIP-event records of the profiled steps: the instantiation with the largest share of the step, and inside it the
    one layer shape that dominates."""
    import csv
    import tempfile
    fd, path = tempfile.mkstemp(suffix=".csv")
    os.close(fd)
    try:
        eng.profile_dump(path)
        rows = [r for r in csv.DictReader(open(path)) if int(r["class"]) == 0]
    finally:
        os.remove(path)
    if not rows:
        return None
    by_k, by_s = {}, {}
    tot_ms = tot_gf = tot_issued = 0.0
    for r in rows:
        for d, key in ((by_k, r["kernel"]), (by_s, (r["kernel"], r["label"]))):
            a = d.setdefault(key, [0, 0.0, 0.0])
            a[0] += 1
            a[1] += float(r["ms"])
            a[2] += float(r["gflop"])
        tot_ms += float(r["ms"])
        tot_gf += float(r["gflop"])
        tot_issued += float(r["gflop"]) * issued_share(r["kernel"])
    kname, (kn, kms, kgf) = max(by_k.items(), key=lambda kv: kv[1][1])
    (_, sname), (sn, sms, sgf) = max(((k, v) for k, v in by_s.items() if k[0] == kname), key=lambda kv: kv[1][1])

    def line(n, ms, gf, share=1.0):
        tf = gf / ms if ms > 0 else 0.0            # GFLOP / ms = TFLOP/s
        d = {"launches_per_step": n // steps, "avg_launch_us": round(ms / n * 1e3, 2),
             "gflop_per_launch": round(gf / n, 3), "ms_per_step": round(ms / steps, 3),
             "achieved": round(tf, 2), "frac": round(tf / PEAK_F32_MFMA, 4)}
        if share != 1.0:
            # Winograd F(2x2,3x3): `achieved` prices the ALGORITHMIC flops of the convolution (2 x 9 x Cin x
            # Cout per pixel); the matrix pipe is issued 4/9 of them -- its own utilisation is this line
            d["mfma_issued_frac"] = round(tf * share / PEAK_F32_MFMA, 4)
        return d
    out = {"kernel": kname, "unit": "TFLOP/s", "peak": PEAK_F32_MFMA}
    out.update(line(kn, kms, kgf, issued_share(kname)))
    out["class_mfma_issued_frac"] = round(tot_issued / tot_ms / PEAK_F32_MFMA, 4) if tot_ms > 0 else None
    out["class_by_kernel"] = {k: {"ms_per_step": round(v[1] / steps, 3), "frac": round(v[2] / v[1] / PEAK_F32_MFMA, 4)}
                              for k, v in sorted(by_k.items(), key=lambda kv: -kv[1][1]) if v[1] > 0}
    out["recompute"] = ("gflop_per_launch / avg_launch_us / %.1f; the average duration of this kernel name in "
                        "profiles/*_kernel_stats.csv (rocprofv3 --kernel-trace --stats over the same command with "
                        "DEPGAN_BENCH_STEP_ONLY=1) must agree with avg_launch_us" % PEAK_F32_MFMA)
    out["largest_shape"] = dict(shape=sname, **line(sn, sms, sgf, issued_share(kname)))
    return out


def bench_config4(dg, torch, dev, B, steps=5):
    """Canonical step of BASELINE configs[3]: DEP-GAN-PROB input (map + FLAIR, nicg = 2), bf16 matrix pipe."""
    x, y2, z, ep = synth(2000, B)
    x = np.concatenate([x, np.roll(x, 7, axis=1)], axis=-1)                 # second input channel (FLAIR stand-in)
    x, y2, z, ep = [torch.from_numpy(np.ascontiguousarray(a)).to(dev) for a in (x, y2, z, ep)]
    nets = [dg.Gen_UNet2D((256, 256, 2), (32, 1), 32, 1, seed=11), dg.Dis_C2D_FCN1((256, 256, 1), seed=12),
            dg.Dis_C2D_FCN1((256, 256, 1), seed=13)]
    tr = dg.build_trainers(*nets, batchSize=B, weights_dtype="bfloat16", activations_dtype="bfloat16", device=dev)

    def step():
        tr.netD_y2_train([y2, x, z, ep])
        tr.netD_dem_train([y2, x, z, ep])
        tr.netG_train([x, y2, z])

    for _ in range(2):
        step()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(steps):
        step()
    torch.cuda.synchronize()
    ms = (time.perf_counter() - t0) / steps * 1e3
    eng = tr.engine
    eng.profile(True)
    eng.profile_reset()
    step()
    c_ms, c_n, c_fl = eng.profile_read(0)
    c_by = eng.profile_read_bytes(0)
    w_ms, _, w_fl = eng.profile_read(1)
    o_ms, _, _ = eng.profile_read(2)
    eng.profile(False)
    eng.profile_reset()
    eng.g_forward(x, z)
    torch.cuda.synchronize()
    t1 = time.perf_counter()
    for _ in range(5):
        eng.g_forward(x, z)
    torch.cuda.synchronize()
    gf = (time.perf_counter() - t1) / 5 * 1e3
    eng.close()
    tf = c_fl / (c_ms * 1e-3) / 1e12 if c_ms > 0 else 0.0
    gbs = c_by / (c_ms * 1e-3) / 1e9 if c_ms > 0 else 0.0
    return {"workload": "BASELINE configs[3]: canonical train step, DEP-GAN-PROB 2-channel input 256x256x2, batch %d, "
                        "bf16 weights and bf16 activations into v_mfma_f32_32x32x16_bf16, fp32 accumulate / masters / "
                        "Adam; the weight-gradient contractions run on the same pipe (operands rounded to bf16 while "
                        "staged, K-major fragments through ds_read_b64_tr_b16)" % B,
            "dtype": "bf16 operands, f32 accumulate", "ms_per_step": round(ms, 3),
            "slices_per_s": round(B / (ms * 1e-3), 1),
            "ms_per_step_by_class": {"conv": round(c_ms, 3), "wgrad": round(w_ms, 3), "other": round(o_ms, 3)},
            "g_forward_ms": round(gf, 3),
            "roofline": {"kernel": "igemm_bf16_kernel + the 16-channel layers left on the fp32 pipe (conv class)",
                         "bound": "hbm", "achieved": round(gbs, 1), "peak": 8000.0, "unit": "GB/s",
                         "frac": round(gbs / 8000.0, 4),
                         "note": "algorithmic bytes (operands once, results once) / HIP-event time; the contraction "
                                 "itself runs at %.1f TFLOP/s = %.3f of the 2500 TFLOP/s dense bf16 peak, i.e. this "
                                 "class sits on the HBM side of the roofline" % (tf, tf / 2500.0)}}


def bench_direct(dg, torch, dev, B, x, y2, z, ep, steps=5):
    """The headline workload with every 3x3 convolution on the direct implicit-GEMM kernel (DEPGAN_WINOGRAD=0, read when
    the context is created): what the Winograd kernel is measured against, same process, same inputs."""
    os.environ["DEPGAN_WINOGRAD"] = "0"
    try:
        nets = [dg.Gen_UNet2D((256, 256, 1), (32, 1), 32, 1, seed=1), dg.Dis_C2D_FCN1((256, 256, 1), seed=2),
                dg.Dis_C2D_FCN1((256, 256, 1), seed=3)]
        tr = dg.build_trainers(*nets, batchSize=B, IM_TRSH=0.178, device=dev)
    finally:
        del os.environ["DEPGAN_WINOGRAD"]

    def step():
        tr.netD_y2_train([y2, x, z, ep])
        tr.netD_dem_train([y2, x, z, ep])
        tr.netG_train([x, y2, z])

    for _ in range(2):
        step()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(steps):
        step()
    torch.cuda.synchronize()
    ms = (time.perf_counter() - t0) / steps * 1e3
    eng = tr.engine
    eng.profile(True)
    eng.profile_reset()
    step()
    c_ms, _, c_fl = eng.profile_read(0)
    eng.profile(False)
    eng.profile_reset()
    eng.g_forward(x, z)
    torch.cuda.synchronize()
    t1 = time.perf_counter()
    for _ in range(5):
        eng.g_forward(x, z)
    torch.cuda.synchronize()
    gf = (time.perf_counter() - t1) / 5 * 1e3
    eng.close()
    tf = c_fl / (c_ms * 1e-3) / 1e12 if c_ms > 0 else 0.0
    return {"note": "DEPGAN_WINOGRAD=0: all convolutions on igemm_conv_kernel (direct implicit GEMM)",
            "ms_per_step": round(ms, 3), "slices_per_s": round(B / (ms * 1e-3), 1),
            "conv_class": {"ms_per_step": round(c_ms, 3), "achieved": round(tf, 2), "frac": round(tf / PEAK_F32_MFMA, 4)},
            "g_forward": {"ms": round(gf, 3), "frac": round(23.513e9 * B / (gf * 1e-3) / 1e12 / PEAK_F32_MFMA, 4)}}


def bench_split(dg, torch, dev, B, x, y2, z, ep, steps=5):
    """The headline workload (canonical step, 256x256x1, fp32 data and weights) with depgan_config.f32_split = 6 and 3."""
    out = {"note": "opt-in (build_trainers(f32_split=6) / DEPGAN_F32_SPLIT=6), NOT the headline: every fp32 operand of the MFMA "
                   "convolutions is split exactly into three bf16 terms (3 x 8 significand bits = fp32's 24) and the six "
                   "largest of the nine cross products run on v_mfma_f32_32x32x16_bf16 with fp32 accumulation; dropped terms "
                   "are below 2^-24 |x||w|.  Operator error against float64 (tests/test_gpu_ops.py): six products 1.0e-7 - "
                   "4.3e-7 mean relative, the native fp32 MFMA kernel 1.3e-7 - 5.0e-7 on the same shapes.  Weight gradients "
                   "stay on the fp32 pipe."}
    for mode in (6, 3):
        nets = [dg.Gen_UNet2D((256, 256, 1), (32, 1), 32, 1, seed=1), dg.Dis_C2D_FCN1((256, 256, 1), seed=2),
                dg.Dis_C2D_FCN1((256, 256, 1), seed=3)]
        tr = dg.build_trainers(*nets, batchSize=B, device=dev, f32_split=mode)

        def step():
            tr.netD_y2_train([y2, x, z, ep])
            tr.netD_dem_train([y2, x, z, ep])
            tr.netG_train([x, y2, z])

        for _ in range(2):
            step()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(steps):
            step()
        torch.cuda.synchronize()
        ms = (time.perf_counter() - t0) / steps * 1e3
        eng = tr.engine
        eng.profile(True)
        eng.profile_reset()
        step()
        c_ms, _, c_fl = eng.profile_read(0)
        w_ms, _, _ = eng.profile_read(1)
        o_ms, _, _ = eng.profile_read(2)
        eng.profile(False)
        eng.profile_reset()
        eng.g_forward(x, z)
        torch.cuda.synchronize()
        t1 = time.perf_counter()
        for _ in range(5):
            eng.g_forward(x, z)
        torch.cuda.synchronize()
        gf = (time.perf_counter() - t1) / 5 * 1e3
        eng.close()
        eq = c_fl / (c_ms * 1e-3) / 1e12 if c_ms > 0 else 0.0
        # this mode runs on the bf16 matrix pipe and is priced against THAT pipe: `mode` bf16 products per fp32 MAC make
        # its peak 2500 / mode TFLOP/s fp32-equivalent (417 for six products), not the fp32 pipe's 157.3
        out["products_%d" % mode] = {
            "ms_per_step": round(ms, 3), "slices_per_s": round(B / (ms * 1e-3), 1),
            "ms_per_step_by_class": {"conv": round(c_ms, 3), "wgrad_fp32": round(w_ms, 3), "other": round(o_ms, 3)},
            "conv_class_tflops_fp32_equivalent": round(eq, 1),
            "roofline": {"bound": "mfma", "pipe": "bf16 (v_mfma_f32_32x32x16_bf16), %d products per fp32 MAC" % mode,
                         "achieved": round(eq, 1), "peak": round(2500.0 / mode, 1), "unit": "TFLOP/s fp32-equivalent",
                         "frac": round(eq / (2500.0 / mode), 4)},
            "g_forward_ms": round(gf, 3)}
    return out


def bench_config5(dg, torch, dev, B, steps=5):
    """One Model.train_on_batch of DEP-UResNet (UT:602-606): phase-1 BatchNorm, Dropout, softmax + categorical CE, Adam."""
    rng = np.random.default_rng(5)
    xs, _, z, _ = synth(3000, B)
    xs = ((xs - xs.mean()) / xs.std()).astype(np.float32)                     # z-scored FLAIR stand-in (UT:511)
    lab = np.eye(4, dtype=np.float32)[rng.integers(0, 4, (B, 256, 256))]      # one-hot (B,256,256,4) (UT:565-568)
    x, z, lab = [torch.from_numpy(np.ascontiguousarray(a)).to(dev) for a in (xs, z, lab)]
    eng = dg.Engine(B, 256, 256, 1, lrG=1e-4, beta1=0.9, beta2=0.999, nc_out=4, device=dev)
    for i in range(2):
        eng.uresnet(x, z, lab, "step", drop_seed=i + 1)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for i in range(steps):
        eng.uresnet(x, z, lab, "step", drop_seed=10 + i)
    torch.cuda.synchronize()
    ms = (time.perf_counter() - t0) / steps * 1e3
    eng.profile(True)
    eng.profile_reset()
    eng.uresnet(x, z, lab, "step", drop_seed=99)
    parts = [eng.profile_read(k) for k in (0, 1, 2)]
    eng.profile(False)
    eng.profile_reset()
    eng.close()
    tf = 70.6e9 * B / (ms * 1e-3) / 1e12
    return {"workload": "BASELINE configs[4]: DEP-UResNet train_on_batch (batch-statistics BatchNorm, Dropout(0.25), softmax + "
                        "categorical cross-entropy, Adam(0.9, 0.999)), batch %d, 256x256x1, fp32" % B,
            "ms_per_step": round(ms, 3), "slices_per_s": round(B / (ms * 1e-3), 1),
            "achieved_tflops": round(tf, 2), "frac_of_fp32_mfma_peak": round(tf / PEAK_F32_MFMA, 4),
            "gflop_per_slice": 70.6,
            "ms_per_step_by_class": {"conv": round(parts[0][0], 3), "wgrad": round(parts[1][0], 3),
                                     "other (batch statistics, affine / dropout passes, noise MLP)": round(parts[2][0], 3)}}


def launch_ranks(args, argv):
    """`python bench.py --gpus N` with no launcher around it: start N ranks as a child torch.distributed.run (this
    process has not touched HIP and never will), relay rank 0's JSON line, exit with the child's status."""
    import socket
    import subprocess
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(args.gpus),
           "--master-addr", "127.0.0.1", "--master-port", str(port), os.path.abspath(__file__)] + argv
    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    env.setdefault("OMP_NUM_THREADS", "4")
    proc = subprocess.Popen(cmd, env=env, stdout=subprocess.PIPE, text=True)
    line = None
    for ln in proc.stdout:
        if ln.startswith("{") and '"metric"' in ln:
            line = ln.strip()
        else:
            sys.stderr.write(ln)
    rc = proc.wait()
    if line is not None:
        print(line, flush=True)
    elif rc == 0:
        rc = 1
        sys.stderr.write("bench.py: the ranks finished without a result line\n")
    return rc


class _DryEngine:
    """Host stand-in for the GPU engine (--dry-run): no arithmetic of the path, only its communication pattern -- a
    gradient arena with the loss pieces in its tail, summed over the ranks once per update through the same
    DataParallel hook the library calls."""

    def __init__(self, dp):
        self.dp, self.device = dp, None
        self.arenas = {"G": np.zeros(2486145 // 64 + 8, np.float32), "D_y2": np.zeros(1798002 // 64 + 8, np.float32),
                       "D_dem": np.zeros(1798002 // 64 + 8, np.float32)}

    def update(self, net, rank):
        a = self.arenas[net]
        a[:] = rank + 1.0
        if self.dp is not None:
            self.dp.allreduce_ptr(a.ctypes.data, a.size)
        time.sleep(0.002)
        return float(a[-1])


def dry_run(args, world, rank):
    import torch.distributed as dist
    dp = None
    if world > 1:
        from dep_gan_im_amd.dist import DataParallel
        dist.init_process_group("gloo")
        dp = DataParallel()
    eng = _DryEngine(dp)

    def step():
        return [eng.update(n, rank) for n in ("D_y2", "D_dem", "G")]

    for _ in range(args.warmup):
        step()
    if dp is not None:
        dist.barrier()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        got = step()
    if dp is not None:
        dist.barrier()
    dt = time.perf_counter() - t0
    want = float(sum(range(1, world + 1)))
    assert all(abs(g - want) < 1e-6 for g in got), (got, want)     # every rank saw the sum over all ranks
    if dp is not None:
        import torch
        t = torch.tensor([dt], dtype=torch.float64)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        dt = float(t.item())
    if rank == 0:
        B = args.batch
        print(json.dumps({"metric": "2D slices/sec (G+2D+GP train step), 256x256x1 fp32", "dry_run": True,
                          "value": round(B * world * args.steps / dt, 3), "unit": "slices/s", "n_gpus": world,
                          "steps": args.steps, "warmup": args.warmup, "ms_per_step": round(dt / args.steps * 1e3, 3),
                          "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": "f32",
                          "data": "none (host stand-in, gloo): rehearsal of the rank plumbing only",
                          "config": {"workload": "dry run", "per_gpu_batch": B, "global_batch": B * world,
                                     "parallelism": "dp%d" % world},
                          "collectives_per_step": 3}), flush=True)
    if dp is not None:
        dist.destroy_process_group()
    return 0


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=10)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--batch", type=int, default=32, help="per-GPU batch (32 = BASELINE config)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--cpu-sample", type=int, default=32)
    ap.add_argument("--dry-run", action="store_true", help="no GPU: host stand-in engine + gloo (plumbing rehearsal)")
    ap.add_argument("--n1-value", type=float, default=None,
                    help="slices/s of the N=1 run, to print scaling_efficiency next to an N>1 value")
    args = ap.parse_args()

    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        return launch_ranks(args, sys.argv[1:])
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    if args.gpus != world:
        sys.stderr.write("bench.py: --gpus %d but WORLD_SIZE=%d: start it as `python bench.py --gpus N` or under "
                         "torch.distributed.run with --nproc-per-node equal to --gpus\n" % (args.gpus, world))
        return 2
    if args.dry_run:
        return dry_run(args, world, rank)

    # The CPU leg runs FIRST, before this process touches the GPU: the GPU phase that follows is then one contiguous
    # stretch at the end of the run instead of a few seconds hidden in front of ~20 s of host work.
    cpu = None
    if world == 1 and rank == 0 and not args.no_cpu_baseline:
        cpu = cpu_baseline(args.cpu_sample)

    import torch
    import dep_gan_im_amd as dg
    from dep_gan_im_amd.build import build
    build()

    # DEPGAN_BENCH_ONE_GPU=1: rehearsal of the multi-rank path on a box with ONE GPU -- every rank uses cuda:0 and the group
    # is gloo (RCCL refuses two ranks on one device); the slices/s of such a run mean nothing, the plumbing is what runs
    one_gpu = bool(os.environ.get("DEPGAN_BENCH_ONE_GPU"))
    if one_gpu:
        local = 0
    torch.cuda.set_device(local)
    dev = torch.device("cuda:%d" % local)
    dp = None
    if world > 1 or os.environ.get("DEPGAN_FORCE_DIST"):   # the env switch rehearses the RCCL path on one GPU
        import torch.distributed as dist
        from dep_gan_im_amd.dist import DataParallel
        os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
        if one_gpu:
            dist.init_process_group("gloo")
        else:
            dist.init_process_group("nccl", device_id=dev)
        # one rank per GPU: the library's own RCCL communicator (ncclAllReduce issued from C on the engine's stream); the
        # process group above carries the 128-byte id, the barrier and the max-over-ranks of the timing
        dp = DataParallel(host_staging=one_gpu)

    B = args.batch
    # deliberately different seeds per rank: build_trainers(dist=dp) must make the replicas identical (rank 0's weights)
    netG = dg.Gen_UNet2D((256, 256, 1), (32, 1), 32, 1, seed=1 + 100 * rank)
    netD1 = dg.Dis_C2D_FCN1((256, 256, 1), seed=2 + 100 * rank)
    netD2 = dg.Dis_C2D_FCN1((256, 256, 1), seed=3 + 100 * rank)
    # the workload is DEP-GAN-IM (irregularity map): IM_TRSH = 0.178 (GT:25-29; 0.5 is the probability-map setting)
    tr = dg.build_trainers(netG, netD1, netD2, batchSize=B, delta=10.0, lrD=1e-4, lrG=1e-4, IM_TRSH=0.178, dist=dp,
                           device=dev)
    x, y2, z, ep = [torch.from_numpy(a).to(dev) for a in synth(1000 + rank, B)]

    def step():
        tr.netD_y2_train([y2, x, z, ep])
        tr.netD_dem_train([y2, x, z, ep])
        tr.netG_train([x, y2, z])

    def barrier():
        if dp is not None:
            torch.distributed.barrier()
        torch.cuda.synchronize()

    for _ in range(args.warmup):
        step()
    barrier()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        step()
    barrier()
    dt = time.perf_counter() - t0
    if dp is not None:
        t = torch.tensor([dt], dtype=torch.float64, device=None if one_gpu else dev)
        torch.distributed.all_reduce(t, op=torch.distributed.ReduceOp.MAX)
        dt = float(t.item())
    ms = dt / args.steps * 1e3
    value = B * world * args.steps / dt
    single_ms = None
    if dp is not None and not os.environ.get("DEPGAN_BENCH_STEP_ONLY"):
        # the same step without its collectives, on an engine of its own (replicas must not diverge), max over ranks
        n1 = [dg.Gen_UNet2D((256, 256, 1), (32, 1), 32, 1, seed=1), dg.Dis_C2D_FCN1((256, 256, 1), seed=2),
              dg.Dis_C2D_FCN1((256, 256, 1), seed=3)]
        t1 = dg.build_trainers(*n1, batchSize=B, IM_TRSH=0.178, device=dev)
        for i in range(2 + args.steps):
            if i == 2:
                torch.cuda.synchronize()
                ts = time.perf_counter()
            t1.netD_y2_train([y2, x, z, ep])
            t1.netD_dem_train([y2, x, z, ep])
            t1.netG_train([x, y2, z])
        torch.cuda.synchronize()
        single_ms = (time.perf_counter() - ts) / args.steps * 1e3
        t1.engine.close()
        t = torch.tensor([single_ms], dtype=torch.float64, device=None if one_gpu else dev)
        torch.distributed.all_reduce(t, op=torch.distributed.ReduceOp.MAX)
        single_ms = float(t.item())

    # ---- roofline of the dominant kernel class, HIP events on the engine's stream ----
    eng = tr.engine
    eng.profile(True)
    eng.profile_reset()
    for _ in range(2):
        step()
    conv_ms, conv_n, conv_fl = eng.profile_read(0)
    conv_bytes = eng.profile_read_bytes(0)
    wg_ms, wg_n, wg_fl = eng.profile_read(1)
    ot_ms, ot_n, _ = eng.profile_read(2)
    if rank == 0 and os.environ.get("DEPGAN_PROFILE_DUMP"):
        eng.profile_dump(os.environ["DEPGAN_PROFILE_DUMP"])     # per-launch labelled CSV (layer shapes) of two steps
    dominant = dominant_kernel(eng, 2) if rank == 0 else None
    eng.profile(False)
    eng.profile_reset()
    achieved = conv_fl / (conv_ms * 1e-3) / 1e12 if conv_ms > 0 else 0.0
    # DEPGAN_BENCH_STEP_ONLY=1: the process runs canonical steps and NOTHING else (no generator-forward / generator-
    # iteration probes, no extra engines) -- the command the rocprofv3 passes of tools/collect_profiles.sh wrap, so that
    # their per-kernel averages and PMC sums cover exactly the launches `roofline` describes
    step_only = bool(os.environ.get("DEPGAN_BENCH_STEP_ONLY"))
    if step_only:
        if rank == 0:
            print(json.dumps({"metric": "2D slices/sec (G+2D+GP train step), 256x256x1 fp32", "step_only": True,
                              "value": round(value, 3), "unit": "slices/s", "n_gpus": world, "steps": args.steps,
                              "warmup": args.warmup, "ms_per_step": round(ms, 3),
                              "canonical_steps_run": args.warmup + args.steps + 2,
                              "conv_class": {"launches_per_step": conv_n // 2, "achieved": round(achieved, 2),
                                             "avg_launch_us": round(conv_ms / max(conv_n, 1) * 1e3, 2),
                                             "algorithmic_bytes_per_launch": round(conv_bytes / max(conv_n, 1))},
                              "dominant_kernel": dominant}), flush=True)
        if dp is not None:
            torch.distributed.destroy_process_group()
        return 0
    # north_star's secondary target: generator forward alone (A1/A12, 23.513 GFLOP/slice) at this batch
    torch.cuda.synchronize()
    out = eng.g_forward(x, z)
    torch.cuda.synchronize()
    t1 = time.perf_counter()
    for _ in range(5):
        out = eng.g_forward(x, z)
    torch.cuda.synchronize()
    gf_ms = (time.perf_counter() - t1) / 5 * 1e3
    gf_tf = 23.513e9 * B / (gf_ms * 1e-3) / 1e12
    # the reference-schedule generator iteration (GT:791-878 steady state): 5 critic-Y2 + 5 critic-DEM
    # updates, the best-of-10 noise search, one G update = 1013 GFLOP per slice of batch; it consumes 5 batches.
    # Timed both ways: closure by closure (12 host synchronisations) and as ONE library call (depgan_gen_iteration).
    zs = torch.randn(10, B, 32, 1, device=dev)
    x5, y5 = x.repeat(5, 1, 1, 1), y2.repeat(5, 1, 1, 1)
    z5, ep5 = z.repeat(5, 1, 1), ep.repeat(5, 1, 1, 1)

    def gen_iteration():
        for _ in range(5):
            tr.netD_y2_train([y2, x, z, ep])
        for _ in range(5):
            tr.netD_dem_train([y2, x, z, ep])
        outs = tr.netG_no_update_many([x, y2, zs])
        best = min(range(10), key=lambda k: outs[k][0])
        tr.netG_train([x, y2, zs[best]])

    def gen_iteration_fused():
        tr.gen_iteration((x5, y5, z5, ep5, 5), (x5, y5, z5, ep5, 5), (x, y2, zs))

    gi = {}
    for name, fn in (("closures", gen_iteration), ("fused", gen_iteration_fused)):
        fn()
        barrier()
        t1 = time.perf_counter()
        for _ in range(2):
            fn()
        barrier()
        gi[name] = (time.perf_counter() - t1) / 2 * 1e3
    gi_ms = gi["fused"]
    # ---- extra, never the headline: BASELINE configs[3] on the bf16 matrix pipe (256x256x2, bf16 weights AND
    # activations into v_mfma_f32_32x32x16_bf16, fp32 accumulate / masters / Adam) -- its own engine, its own roofline
    config4 = None
    if world == 1 and rank == 0 and not os.environ.get("DEPGAN_BENCH_SKIP_CONFIG4"):
        config4 = bench_config4(dg, torch, dev, B)
    # ---- extra, opt-in mode, never the headline: the SAME fp32 workload with the convolutions' fp32 operands split exactly
    # into bf16 terms and the six largest cross products on the bf16 matrix pipe (depgan_config.f32_split = 6) ----
    f32_split = None
    if world == 1 and rank == 0 and not os.environ.get("DEPGAN_BENCH_SKIP_SPLIT") and not tr.engine.f32_split:
        f32_split = bench_split(dg, torch, dev, B, x, y2, z, ep)
    direct_conv = None
    if world == 1 and rank == 0 and not os.environ.get("DEPGAN_BENCH_SKIP_DIRECT") and os.environ.get("DEPGAN_WINOGRAD") != "0":
        direct_conv = bench_direct(dg, torch, dev, B, x, y2, z, ep)
    # ---- extra: BASELINE configs[4], the DEP-UResNet supervised step (learning phase 1, softmax / CE head) ----
    config5 = None
    if world == 1 and rank == 0 and not os.environ.get("DEPGAN_BENCH_SKIP_CONFIG5"):
        config5 = bench_config5(dg, torch, dev, B)
    traffic, traffic_src = pmc_traffic(B, conv_n // 2)
    roofline = {"bound": "mfma", "achieved": round(achieved, 2), "peak": PEAK_F32_MFMA, "unit": "TFLOP/s",
                "frac": round(achieved / PEAK_F32_MFMA, 4), "traffic": traffic,
                "traffic_unit": "HBM bytes per launch (rocprofv3 FETCH_SIZE x2 + WRITE_SIZE, separate --pmc passes; "
                                "%s)" % traffic_src,
                "algorithmic_bytes_per_launch": round(conv_bytes / max(conv_n, 1)),
                "kernel": "the convolution class: wino_conv_kernel (3x3 layers of 32-channel tiles: Winograd F(2x2,3x3) on "
                          "the fp32 matrix pipe, 4/9 of the direct form's MFMAs) + igemm_conv_kernel (5x5, 1x1, 16-channel "
                          "tiles, GP u-forward) + the 3 deconv_fwd_kernel launches of the generator forward; `achieved` "
                          "prices ALGORITHMIC flops (SURVEY 8d), dominant_kernel.class_mfma_issued_frac the flops the "
                          "matrix pipe was actually issued; direct_conv = the same step with DEPGAN_WINOGRAD=0",
                "avg_launch_us": round(conv_ms / max(conv_n, 1) * 1e3, 2), "launches_per_step": conv_n // 2,
                "dominant_kernel": dominant,
                "wgrad": {"achieved": round(wg_fl / (wg_ms * 1e-3) / 1e12, 2) if wg_ms > 0 else 0.0,
                          "ms_per_step": round(wg_ms / 2, 3)},
                "ms_per_step": {"igemm_conv": round(conv_ms / 2, 3), "wgrad": round(wg_ms / 2, 3),
                                "other": round(ot_ms / 2, 3)},
                "whole_step_frac": round(GFLOP_PER_SLICE * 1e9 * B / (ms * 1e-3) / 1e12 / PEAK_F32_MFMA, 4),
                "gen_iteration": {"ms": round(gi_ms, 2), "ms_closure_by_closure": round(gi["closures"], 2),
                                  "host_syncs": 1, "gflop_per_slice_of_batch": 1013.0,
                                  "achieved": round(1013.0e9 * B / (gi_ms * 1e-3) / 1e12, 2),
                                  "epoch_slices_per_s": round(5 * B * world / (gi_ms * 1e-3), 1),
                                  "unit": "5 critic-Y2 + 5 critic-DEM + best-of-10 + 1 G update (GT:791-878), one "
                                          "library call / one host synchronisation"},
                "g_forward": {"ms": round(gf_ms, 3), "achieved": round(gf_tf, 2),
                              "frac": round(gf_tf / PEAK_F32_MFMA, 4), "slices_per_s": round(B / (gf_ms * 1e-3), 1)}}

    if rank == 0:
        line = {"metric": "2D slices/sec (G+2D+GP train step), 256x256x1 fp32", "value": round(value, 3),
                "unit": "slices/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
                "ms_per_step": round(ms, 3), "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
                "dtype": "f32", "data": "synthetic" if not one_gpu else "synthetic; REHEARSAL: all ranks on one GPU, gloo",
                "config": {"workload": "DEP-GAN-IM twoCritics canonical train step (critic-Y2 + critic-DEM + G update, "
                                       "WGAN-GP), batch %d per GPU, 256x256x1" % B,
                           "per_gpu_batch": B, "global_batch": B * world, "parallelism": "dp%d" % world},
                "roofline": roofline}
        if dp is not None:
            line["collectives"] = {"per_step": 3, "issued": dp.calls,
                                   "expected": 3 * (args.warmup + args.steps + 2) + 2 * 12 * 3,
                                   "path": "direct RCCL: ncclAllReduce from libdepgan on the engine's stream" if dp.direct
                                           else "torch.distributed hook (%s)" % ("host staging, gloo" if one_gpu else "nccl"),
                                   "rccl_nranks": eng.rccl_info()[0] if dp.direct else None,
                                   "message_floats": [int(eng.arena(n, 2)[1]) + 8 for n in ("D_y2", "D_dem", "G")]}
            if single_ms is not None:
                # self-check for the first multi-GPU run: the same step on the same GPU without the collectives (a fresh
                # single-rank engine, max over ranks) -- weak-scaling efficiency is its ratio to the timed step
                line["collectives"]["ms_per_step_without_collectives"] = round(single_ms, 3)
                line["scaling_efficiency"] = round(single_ms / ms, 4)
        if args.n1_value:
            line["scaling_efficiency_vs_n1_value"] = round(value / (world * args.n1_value), 4)
        if config4 is not None:
            line["config4"] = config4
        if config5 is not None:
            line["config5"] = config5
        if f32_split is not None:
            line["f32_split"] = f32_split
        if direct_conv is not None:
            line["direct_conv"] = direct_conv
        if cpu is not None:
            line["cpu_baseline"] = cpu
        print(json.dumps(line), flush=True)
    if dp is not None:
        torch.distributed.destroy_process_group()
    return 0


if __name__ == "__main__":
    sys.exit(main())
