"""Batch-sharded data parallelism for the train step.

The reference is single-GPU (GT:13).  Here every rank holds full replicas of G
and both critics and a shard of the batch; per network update there is exactly
one collective on the flat fp32 gradient arena (RCCL all-reduce over xGMI when
the process group backend is "nccl"; gloo on CPU in the tests), followed by a
tiny all-reduce of un-normalised loss pieces so that every rank reports the
same global scalars and -- because M3/M4 (GT:583-589) are non-linear in the
batch-global counts -- the same best-of-k noise choice (GT:868-877).
"""
from __future__ import annotations

import torch
import torch.distributed as dist


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
    -> the six scalars of netG_no_update / netG_train (GT:576-598)."""
    n, npix = float(s[6]), float(s[7])
    lf, lfd = float(s[0]) / n, float(s[1]) / n
    m1 = 100.0 * float(s[2]) / npix
    dv = float(s[3]) / 1000.0 - float(s[4]) / 1000.0
    m3 = 100.0 * dv * dv
    dice = (2.0 * float(s[5]) + 1e-7) / (float(s[3]) + float(s[4]) + 1e-7)
    m4 = 1.0 - dice
    return [-lf - lfd + m1 + m3 + m4, lf, lfd, m1, m3, m4]


class DataParallel:
    def __init__(self, group=None):
        if not dist.is_initialized():
            raise RuntimeError("torch.distributed must be initialised before building DataParallel")
        self.group = group
        self.world = dist.get_world_size(group)
        self.rank = dist.get_rank(group)
        self._views = {}

    def _grad_tensor(self, engine, net):
        if hasattr(engine, "grad_tensor"):          # test doubles / CPU engines
            return engine.grad_tensor(net)
        key = (id(engine), net)
        if key not in self._views:
            ptr, n = engine.grad_arena(net)
            self._views[key] = torch.as_tensor(_DevArray(ptr, n), device=engine.device)
        return self._views[key]

    def _allreduce_grads(self, engine, net):
        # losses are batch means (GT:540-545, 576): the global gradient is the mean of the rank gradients
        g = self._grad_tensor(engine, net)
        if dist.get_backend(self.group) == "nccl":
            dist.all_reduce(g, op=dist.ReduceOp.AVG, group=self.group)    # RCCL averages inside the collective
        else:                                                            # gloo (CPU tests) has no AVG
            dist.all_reduce(g, op=dist.ReduceOp.SUM, group=self.group)
            g.mul_(1.0 / self.world)

    def _allreduce_sums(self, engine, n):
        s = torch.tensor(engine.last_sums()[:n], dtype=torch.float64)
        dev = getattr(engine, "device", None)
        if dev is not None and dist.get_backend(self.group) == "nccl":
            s = s.to(dev)
        dist.all_reduce(s, op=dist.ReduceOp.SUM, group=self.group)
        return s.cpu().tolist()

    def reduce_critic(self, engine, which, out):
        self._allreduce_grads(engine, which)
        return combine_critic_sums(self._allreduce_sums(engine, 4))

    def reduce_generator_many(self, sums, device=None):
        """k x 8 un-normalised pieces -> k x 6 global scalars with ONE all-reduce (same choice on every rank)."""
        s = torch.tensor(sums, dtype=torch.float64)
        if device is not None and dist.get_backend(self.group) == "nccl":
            s = s.to(device)
        dist.all_reduce(s, op=dist.ReduceOp.SUM, group=self.group)
        return [combine_generator_sums(row) for row in s.cpu().tolist()]

    def reduce_generator(self, engine, out, grads):
        if grads:
            self._allreduce_grads(engine, "G")
        return combine_generator_sums(self._allreduce_sums(engine, 8))
