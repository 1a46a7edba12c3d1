"""CPU, world_size 2, gloo: the data-parallel reduction logic --
gradient arenas averaged, loss scalars recomputed from global sums so that
M3/M4 (non-linear in batch-global counts) match the single-process value."""
import os
import socket

import numpy as np
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from dep_gan_im_amd.dist import DataParallel, combine_critic_sums, combine_generator_sums


class FakeEngine:
    def __init__(self, grads, sums):
        self._g = {k: torch.tensor(v, dtype=torch.float32) for k, v in grads.items()}
        self._s = list(sums) + [0.0] * (8 - len(sums))

    def grad_tensor(self, net):
        return self._g[net]

    def last_sums(self):
        return self._s


def _sums_for(rank):
    rng = np.random.default_rng(100 + rank)
    n, npix = 4.0, 4.0 * 64
    return [rng.normal(), rng.normal(), abs(rng.normal()) * 50, float(rng.integers(5, 40)), float(rng.integers(5, 40)),
            float(rng.integers(0, 5)), n, npix]


def _worker(rank, world, port, q):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    dp = DataParallel()
    g = {"G": np.full(10, float(rank + 1)), "D_y2": np.arange(6.0) * (rank + 1)}
    eng = FakeEngine(g, _sums_for(rank))
    out_g = dp.reduce_generator(eng, None, grads=True)
    eng2 = FakeEngine(g, _sums_for(rank)[:2] + [1.5, 4.0])
    out_c = dp.reduce_critic(eng2, "D_y2", None)
    many = dp.reduce_generator_many([_sums_for(10 * k + rank) for k in range(3)])
    q.put((rank, out_g, eng.grad_tensor("G").tolist(), out_c, eng2.grad_tensor("D_y2").tolist(), many))
    dist.destroy_process_group()


def test_two_rank_reduction_matches_single_process():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    procs = [ctx.Process(target=_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    res = sorted(q.get(timeout=120) for _ in range(2))
    for p in procs:
        p.join(60)
        assert p.exitcode == 0
    tot = [a + b for a, b in zip(_sums_for(0), _sums_for(1))]
    want_g = combine_generator_sums(tot)
    want_many = [combine_generator_sums([a + b for a, b in zip(_sums_for(10 * k), _sums_for(10 * k + 1))])
                 for k in range(3)]
    for rank, out_g, gG, out_c, gD, many in res:
        np.testing.assert_allclose(many, want_many, rtol=1e-12)     # best-of-k: one all-reduce, same on all ranks
        np.testing.assert_allclose(out_g, want_g, rtol=1e-12)
        np.testing.assert_allclose(gG, np.full(10, 1.5), rtol=1e-6)          # mean of 1 and 2
        np.testing.assert_allclose(gD, np.arange(6.0) * 1.5, rtol=1e-6)
        np.testing.assert_allclose(out_c, combine_critic_sums([tot[0], tot[1], 3.0, 8.0]), rtol=1e-12)
    assert res[0][1] == res[1][1]   # identical scalars on both ranks -> identical arg-min noise


def test_m3_m4_are_not_rank_averages():
    a, b = _sums_for(0), _sums_for(1)
    tot = [x + y for x, y in zip(a, b)]
    glob = combine_generator_sums(tot)
    avg = [(x + y) / 2 for x, y in zip(combine_generator_sums(a), combine_generator_sums(b))]
    assert abs(glob[4] - avg[4]) > 1e-9      # M3: square of a global count difference
    assert abs(glob[1] - avg[1]) < 1e-12     # plain batch means do average
