"""Batch-sharded data parallelism for the train step.

The reference is single-GPU (GT:13).  Here every rank holds full replicas of G
and both critics and a shard of the batch.  On GPUs the library talks to RCCL
itself (include/depgan.h, depgan_rccl_*): this module only hands rank 0's
ncclUniqueId to the other ranks (one torch.distributed broadcast of 128 bytes,
any backend) and asks the library to broadcast rank 0's weights; from then on
every update's collective is an ncclAllReduce the library enqueues on its own
stream -- no Python, no GIL on the data path.  The hook form
(depgan_set_allreduce, implemented here with torch.distributed) remains for
host memory (gloo: CPU tests with an engine double) and for two ranks sharing
one GPU (tests / DEPGAN_BENCH_ONE_GPU rehearsal, where RCCL refuses to run),
and must be asked for explicitly when the engine is on a GPU (host_staging=True):
it synchronises the stream on every collective.  Per network update there is
exactly one collective: the flat fp32 gradient arena with the
update's un-normalised loss pieces riding in its tail, summed; Adam divides the
gradient by the world size (losses are batch means, GT:540-545, 576), and every
rank forms the same GLOBAL scalars from the summed pieces -- M3/M4 (GT:583-589)
are non-linear in the batch-global counts -- hence the same best-of-k noise
choice (GT:868-877), which the library takes on the device.  Nothing here
synchronises the host: a whole generator iteration (depgan_gen_iteration) is
one enqueue with its collectives in stream order.
"""
from __future__ import annotations

import ctypes as C

import numpy as np
import torch
import torch.distributed as dist

from ._lib import ARENA_ADAM_M, ARENA_ADAM_V, ARENA_NONTRAINABLE, ARENA_PARAMS


class _DevArray:
    """__cuda_array_interface__ view of a raw device pointer (no copy)."""

    def __init__(self, ptr, n):
        self.__cuda_array_interface__ = {"shape": (int(n),), "typestr": "<f4", "data": (int(ptr), False), "version": 2}


def combine_critic_sums(s):
    """s = [sum D(real), sum D(fake), sum (norm-1)^2, n] -> [loss_real, loss_fake] (GT:540-541)."""
    n = float(s[3])
    return [float(s[0]) / n, float(s[1]) / n]


def combine_generator_sums(s):
    """s = [sum D_y2(fake), sum D_dem(attr), sum|attr-real_dem|, sum wr, sum wf, sum wr*wf, n, n*H*W]
    -> the six scalars of netG_no_update / netG_train (GT:576-598).  The library does the same algebra
    (g_loss_from_sums in csrc/model.hip); this copy serves the tests and external callers of *_grads."""
    n, npix = float(s[6]), float(s[7])
    lf, lfd = float(s[0]) / n, float(s[1]) / n
    m1 = 100.0 * float(s[2]) / npix
    dv = float(s[3]) / 1000.0 - float(s[4]) / 1000.0
    m3 = 100.0 * dv * dv
    dice = (2.0 * float(s[5]) + 1e-7) / (float(s[3]) + float(s[4]) + 1e-7)
    m4 = 1.0 - dice
    return [-lf - lfd + m1 + m3 + m4, lf, lfd, m1, m3, m4]


class DataParallel:
    def __init__(self, group=None, host_staging=False, direct_rccl=True):
        """host_staging: allow a GPU engine under a group whose collectives run on host memory (gloo): every message is
        copied through the host and the stream is synchronised -- tests and rehearsals only.
        direct_rccl: GPU engines use the library's own RCCL communicator (the default); False keeps the
        torch.distributed hook on an "nccl" group (the round-2 path, kept for comparison)."""
        if not dist.is_initialized():
            raise RuntimeError("torch.distributed must be initialised before building DataParallel")
        self.group = group
        self.world = dist.get_world_size(group)
        self.rank = dist.get_rank(group)
        backend = str(dist.get_backend(group))
        # a composite backend ("cpu:gloo,cuda:nccl") serves device tensors through its nccl part
        self.on_device = "nccl" in backend
        self.host_staging = bool(host_staging)
        self.direct_rccl = bool(direct_rccl)
        self.direct = False      # set by attach(): the engine reduces through its own RCCL communicator
        self._views = {}
        self.device = None
        self.engine = None
        self._calls = 0          # collectives issued through the hook (tests / bench report `calls`)

    @property
    def calls(self):
        """Collectives issued so far: by the library's RCCL communicator (as it counts them) or through the hook."""
        if self.direct and self.engine is not None:
            return self.engine.rccl_info()[2]
        return self._calls

    # ---- the hook: in-place summing all-reduce of n floats at a raw pointer ----
    def _view(self, ptr, n):
        key = (ptr, n)
        t = self._views.get(key)
        if t is None:
            if self.on_device:
                t = torch.as_tensor(_DevArray(ptr, n), device=self.device)
            else:                                   # gloo: host memory (CPU tests with an engine double)
                t = torch.from_numpy(np.ctypeslib.as_array(C.cast(ptr, C.POINTER(C.c_float)), shape=(n,)))
            self._views[key] = t
        return t

    def allreduce_ptr(self, ptr, n, stream=0):
        self._calls += 1
        if not self.on_device and self.device is not None and getattr(self.device, "type", "cpu") == "cuda":
            # gloo group around GPU engines (tests: two ranks sharing one GPU, where RCCL refuses to run): the message is
            # staged through the host.  Synchronises the stream -- a test path, not the product's.
            key = ("dev", ptr, n)
            t = self._views.get(key)
            if t is None:
                t = self._views[key] = torch.as_tensor(_DevArray(ptr, n), device=self.device)
            torch.cuda.current_stream(self.device).synchronize()
            h = t.cpu()
            dist.all_reduce(h, op=dist.ReduceOp.SUM, group=self.group)
            t.copy_(h)
            return
        t = self._view(ptr, n)
        if self.on_device:
            cur = torch.cuda.current_stream(self.device)
            if stream and cur.cuda_stream != stream:    # the engine was put on another stream: order against THAT one
                with torch.cuda.stream(torch.cuda.ExternalStream(stream, device=self.device)):
                    dist.all_reduce(t, op=dist.ReduceOp.SUM, group=self.group)
                return
        dist.all_reduce(t, op=dist.ReduceOp.SUM, group=self.group)

    # ---- engine attachment ----
    def attach(self, engine, nets=("G", "D_y2", "D_dem")):
        """Makes `engine` one replica of a data-parallel job: rank 0's weights, BN moving statistics and optimiser
        state replace every other rank's (replicas built from different seeds would otherwise apply the averaged
        gradient to different models and never agree), then the all-reduce hook is registered."""
        self.device = getattr(engine, "device", None)
        self.engine = engine
        on_gpu = getattr(self.device, "type", "cpu") == "cuda"
        if on_gpu and self.direct_rccl and not self.host_staging:
            return self._attach_direct(engine, nets)
        if on_gpu and not self.on_device and not self.host_staging:
            raise RuntimeError("a GPU engine under a %r process group would stage every collective through the host and "
                               "synchronise the stream: build DataParallel(host_staging=True) if that is what you want "
                               "(tests / one-GPU rehearsals), or leave direct_rccl on" % str(dist.get_backend(self.group)))
        for net in nets:
            for arena in (ARENA_PARAMS, ARENA_NONTRAINABLE, ARENA_ADAM_M, ARENA_ADAM_V):
                ptr, n = engine.arena(net, arena)
                if n:
                    self._broadcast_ptr(ptr, n)
            t = torch.tensor([engine.adam_step(net)], dtype=torch.int64,
                             device=self.device if self.on_device else None)
            dist.broadcast(t, src=self._global_rank0(), group=self.group)
            engine.adam_step(net, int(t.item()))
            engine.weights_changed(net)
        engine.set_allreduce(self.allreduce_ptr, self.world)
        return engine

    def _attach_direct(self, engine, nets):
        """The product path: the library's own RCCL communicator.  torch.distributed carries the 128-byte id only."""
        box = [engine.rccl_unique_id() if self.rank == 0 else None]
        dist.broadcast_object_list(box, src=self._global_rank0(), group=self.group)
        # RCCL prints a version banner on file descriptor 1 when a communicator is created; callers (bench.py) own stdout
        # for their result line, so the banner is sent to stderr
        import os
        import sys
        sys.stdout.flush()
        saved = os.dup(1)
        try:
            os.dup2(2, 1)
            engine.rccl_init(box[0], self.rank, self.world)
        finally:
            os.dup2(saved, 1)
            os.close(saved)
        nranks, rank, _ = engine.rccl_info()
        if (nranks, rank) != (self.world, self.rank):
            raise RuntimeError("RCCL reports rank %d of %d, the process group says %d of %d" % (rank, nranks, self.rank,
                                                                                                self.world))
        steps = []
        for net in nets:
            for arena in (ARENA_PARAMS, ARENA_NONTRAINABLE, ARENA_ADAM_M, ARENA_ADAM_V):
                ptr, n = engine.arena(net, arena)
                if n:
                    engine.rccl_broadcast(ptr, n, 0)
            steps.append(engine.adam_step(net))
        box = [steps if self.rank == 0 else None]
        dist.broadcast_object_list(box, src=self._global_rank0(), group=self.group)
        for net, t in zip(nets, box[0]):
            engine.adam_step(net, int(t))
            engine.weights_changed(net)
        self.direct = True
        return engine

    def _broadcast_ptr(self, ptr, n):
        if not self.on_device and self.device is not None and getattr(self.device, "type", "cpu") == "cuda":
            t = torch.as_tensor(_DevArray(ptr, n), device=self.device)       # gloo around GPU engines: via the host
            torch.cuda.synchronize(self.device)
            h = t.cpu()
            dist.broadcast(h, src=self._global_rank0(), group=self.group)
            t.copy_(h)
            return
        dist.broadcast(self._view(ptr, n), src=self._global_rank0(), group=self.group)

    def _global_rank0(self):
        return dist.get_global_rank(self.group, 0) if self.group is not None else 0

    def shard(self, n_global):
        """Sample range [lo, hi) of this rank inside a global batch of n_global samples (sharded by sample
        index, equal shards)."""
        if n_global % self.world:
            raise ValueError("global batch %d is not divisible by the world size %d" % (n_global, self.world))
        per = n_global // self.world
        return self.rank * per, (self.rank + 1) * per
