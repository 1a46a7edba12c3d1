#!/usr/bin/env python3
"""The training loop of DEP-GAN_PROB_IM_twoCritics_training_4fold.py (GT:779-894) on synthetic slices.

What the reference script does after loading its NIfTI data, with the Keras / TensorFlow lines replaced as
INTEGRATION.md shows: build the three models, build the four closures, then per epoch run the schedule (critic
iterations, best-of-10 noise, one generator update), log, validate with predict, save the generator.

    python examples/train_synthetic.py --epochs 1 --slices 64 --batch 16
    python -m torch.distributed.run --nproc-per-node 2 --master-addr 127.0.0.1 examples/train_synthetic.py   # data parallel

The training set is moved to HBM once (a 288-GB device holds any realistic set of 256x256 slices), every generator
iteration is one library call with one host synchronisation (depgan_gen_iteration), the full training state is
checkpointed each epoch -- networks, Adam state, schedule counters AND the driver's own state: the RNG, the cumulative
data order (the reference reshuffles the already shuffled arrays, GT:783-787) and the epoch -- so --resume continues the
uninterrupted run bit for bit.  Every rank reads the state file rank 0 wrote: one node / a shared filesystem.
"""
import argparse
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))


def synthetic_slices(n, size, seed):
    """x: baseline maps in [0,1] inside an elliptical mask, y2: follow-up maps (blobs grown / shrunk)."""
    rng = np.random.default_rng(seed)
    yy, xx = np.mgrid[0:size, 0:size].astype(np.float64)
    c = (size - 1) / 2.0
    mask = (((yy - c) / (0.39 * size)) ** 2 + ((xx - c) / (0.31 * size)) ** 2) <= 1.0
    x = np.zeros((n, size, size)); y = np.zeros((n, size, size))
    for i in range(n):
        for _ in range(int(rng.integers(3, 9))):
            cy, cx = c + rng.uniform(-0.27, 0.27) * size, c + rng.uniform(-0.21, 0.21) * size
            amp, sig = rng.uniform(0.3, 1.0), rng.uniform(2.0, 8.0) * size / 256
            d2 = (yy - cy) ** 2 + (xx - cx) ** 2
            x[i] += amp * np.exp(-d2 / (2 * sig ** 2))
            y[i] += amp * rng.uniform(0.7, 1.3) * np.exp(-d2 / (2 * (sig * rng.uniform(0.7, 1.3)) ** 2))
        x[i] = np.clip(x[i], 0, 1) * mask
        y[i] = np.clip(y[i], 0, 1) * mask
    return x[..., None].astype(np.float32), y[..., None].astype(np.float32)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--epochs", type=int, default=1)
    ap.add_argument("--slices", type=int, default=64)
    ap.add_argument("--batch", type=int, default=16)       # batchSize, GT:32
    ap.add_argument("--size", type=int, default=256)       # imageSize
    ap.add_argument("--out", default="netG_synthetic.npz")
    ap.add_argument("--state", default="train_state.npz", help="full training state (3 networks, Adam, counters)")
    ap.add_argument("--resume", action="store_true")
    args = ap.parse_args()

    import torch
    import dep_gan_im_amd as dg
    from dep_gan_im_amd.schedule import ScheduleState, train_epoch

    rank, world = int(os.environ.get("RANK", "0")), int(os.environ.get("WORLD_SIZE", "1"))
    torch.cuda.set_device(int(os.environ.get("LOCAL_RANK", "0")))
    dp = None
    if world > 1:                                     # one process per GPU; batchSize is the per-GPU batch
        import torch.distributed as dist
        from dep_gan_im_amd.dist import DataParallel
        dist.init_process_group("nccl", device_id=torch.device("cuda", torch.cuda.current_device()))
        dp = DataParallel()

    imageSize, noiseSize, first_fm_G, nicg = args.size, 32, 32, 1
    netD_y2 = dg.Dis_C2D_FCN1((imageSize, imageSize, 1), seed=1)              # GT:513
    netD_dem = dg.Dis_C2D_FCN1((imageSize, imageSize, 1), seed=2)             # GT:516
    netG = dg.Gen_UNet2D((imageSize, imageSize, nicg), (noiseSize, 1), first_fm_G, 1, seed=3)   # GT:520
    t = dg.build_trainers(netG, netD_y2, netD_dem, batchSize=args.batch, delta=10.0, lrD=1e-4, lrG=1e-4, IM_TRSH=0.178,
                          dist=dp)

    train_1tp, train_2tp = synthetic_slices(args.slices, imageSize, 0)
    train_1tp, train_2tp = torch.from_numpy(train_1tp).cuda(), torch.from_numpy(train_2tp).cuda()   # resident in HBM
    val_1tp, val_2tp = synthetic_slices(max(args.batch // 2, 2), imageSize, 1)
    fixed_noise = np.random.normal(size=(len(val_1tp), noiseSize, 1)).astype("float32")           # GT:772

    state = ScheduleState()
    state.gen_iterations = 26          # skip the 100-iteration critic warm-up of the first 25 iterations (GT:795)
    rng = np.random.RandomState(1234)                             # identical on every rank: draws are for the GLOBAL batch
    order = np.arange(train_1tp.shape[0])                          # cumulative permutation of the training set
    first_epoch = 0
    if args.resume and os.path.exists(args.state):
        t.load_state(args.state, state)
        t.rng_from_arrays(rng, t.extra)
        order, first_epoch = np.asarray(t.extra["order"]), int(t.extra["epoch"]) + 1
        idx = torch.from_numpy(order).cuda()
        train_1tp, train_2tp = train_1tp[idx], train_2tp[idx]
        print("resumed after epoch %d at generator iteration %d" % (first_epoch, state.gen_iterations))

    def log(r):
        if rank != 0:
            return
        print("[%d] D_y2 %.4f (real %.4f fake %.4f)  D_dem %.4f  G %.4f (CY2 %.4f DEM %.4f L1 %.4f VOL %.4f DSC %.4f)  "
              "best noise %d" % (r["gen_iterations"], r["errD"], r["errD_real"], r["errD_fake"], r["errD_dem"], r["errG"],
                                 r["errG_CY2"], r["errG_DEM"], r["errG_MSE"], r["errG_VOL"], r["errG_WMH"], r["best_noise"]),
              flush=True)

    for epoch in range(first_epoch, first_epoch + args.epochs):
        t0 = time.time()
        indices = np.arange(train_1tp.shape[0])                    # GT:783-787: reshuffle the (already shuffled) arrays
        rng.shuffle(indices)
        order = order[indices]
        idx = torch.from_numpy(indices).cuda()
        train_1tp, train_2tp = train_1tp[idx], train_2tp[idx]
        train_epoch(t, train_1tp, train_2tp, batchSize=args.batch, Diters=5, state=state, on_gen_iteration=log, rng=rng,
                    rank=rank, world=world, shuffle=False)
        fake_dem = netG.predict([val_1tp, fixed_noise])                                            # GT:846-859
        val_real = float(netD_y2.predict(val_2tp).mean())
        val_fake = float(netD_y2.predict(val_1tp[..., 0:1] + fake_dem).mean())
        print("epoch %d: %.1f s, %d generator iterations so far; validation D_y2(real) %.4f D_y2(fake) %.4f"
              % (epoch + 1, time.time() - t0, state.gen_iterations, val_real, val_fake), flush=True)
        if rank == 0:
            netG.save(args.out)                                                                    # GT:892
            t.save_state(args.state, state, extra=dict(t.rng_to_arrays(rng), order=order, epoch=epoch))
    if rank == 0:
        print("saved", args.out, "and", args.state)
    if dp is not None:
        torch.distributed.destroy_process_group()


if __name__ == "__main__":
    main()
