"""CPU, world_size 2, gloo: the data-parallel plumbing.

libdepgan calls ONE hook per network update -- an in-place summing all-reduce of the gradient arena with the
update's un-normalised loss pieces in its tail -- then divides the gradient by the world size inside Adam and forms
GLOBAL scalars from the summed pieces.  Here an engine double written in NumPy follows exactly that protocol through
the real dep_gan_im_amd.dist.DataParallel (gloo instead of RCCL), so that the hook, the weight broadcast of attach(),
the rank-sharded schedule and the `python bench.py --gpus N` launcher are exercised without a GPU."""
import json
import os
import socket
import subprocess
import sys

import numpy as np
import torch.distributed as dist
import torch.multiprocessing as mp

from dep_gan_im_amd import _lib
from dep_gan_im_amd.dist import DataParallel, combine_critic_sums, combine_generator_sums
from dep_gan_im_amd.schedule import ScheduleState, train_epoch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


class NumpyEngine:
    """Engine double: arenas are NumPy arrays, 'gradients' a deterministic function of the inputs, the update protocol
    is the library's (csrc/model.hip finish_update): grads + 8 tail floats -> hook (sum) -> Adam with g / world."""
    NETS = ("G", "D_y2", "D_dem")

    def __init__(self, seed, n=12):
        rng = np.random.default_rng(seed)
        self.P = {k: rng.normal(size=n).astype(np.float32) for k in self.NETS}
        self.NT = {k: rng.normal(size=4).astype(np.float32) for k in self.NETS}
        self.M = {k: np.zeros(n, np.float32) for k in self.NETS}
        self.V = {k: np.zeros(n, np.float32) for k in self.NETS}
        self.G = {k: np.zeros(n + 8, np.float32) for k in self.NETS}
        self.t = {k: 0 for k in self.NETS}
        self.device, self.fn, self.world, self.changed = None, None, 1, []

    def arena(self, net, arena):
        a = {_lib.ARENA_PARAMS: self.P, _lib.ARENA_NONTRAINABLE: self.NT, _lib.ARENA_ADAM_M: self.M,
             _lib.ARENA_ADAM_V: self.V}[arena][net]
        return a.ctypes.data, a.size

    def adam_step(self, net, value=None):
        if value is not None:
            self.t[net] = value
        return self.t[net]

    def weights_changed(self, net):
        self.changed.append(net)

    def set_allreduce(self, fn, world):
        self.fn, self.world = fn, world

    def update(self, net, data, stats):
        g = self.G[net]
        g[:-8] = np.float32(data.mean()) * np.arange(1, g.size - 7, dtype=np.float32)   # linear in the shard mean
        g[-8:] = 0
        n0 = g.size - 8
        g[n0:n0 + len(stats)] = stats
        if self.fn is not None:
            self.fn(g.ctypes.data, g.size, 0)
        self.P[net] -= np.float32(0.1) * g[:-8] / np.float32(self.world)
        self.t[net] += 1
        return g[-8:].copy()


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
    eng = NumpyEngine(seed=rank)                     # replicas start DIFFERENT (seed=None per rank)
    eng.M["G"][:] = rank
    eng.t["D_dem"] = 5 * rank + 2
    dp.attach(eng)
    after_attach = {k: eng.P[k].copy() for k in eng.NETS}
    nt = {k: eng.NT[k].copy() for k in eng.NETS}
    data = np.full((4, 2), float(rank + 1), np.float32)            # this rank's shard of the global batch
    sg = eng.update("G", data, np.float32(_sums_for(rank)))
    sc = eng.update("D_y2", data, np.float32(_sums_for(rank)[:2] + [1.5, 4.0]))
    q.put((rank, after_attach, nt, eng.M["G"].copy(), eng.t["D_dem"], sorted(set(eng.changed)), sg, sc,
           {k: eng.P[k].copy() for k in eng.NETS}, dp.calls))
    dist.destroy_process_group()


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    return port


def _run_two(target):
    port = _free_port()
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    procs = [ctx.Process(target=target, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    res = sorted((q.get(timeout=60) for _ in range(2)), key=lambda t: t[0])
    for p in procs:
        p.join(60)
        assert p.exitcode == 0
    return res


def test_attach_broadcasts_rank0_and_updates_are_global():
    r0, r1 = _run_two(_worker)
    ref = NumpyEngine(seed=0)
    for k in NumpyEngine.NETS:                       # attach(): rank 0's weights and BN statistics everywhere
        np.testing.assert_array_equal(r0[1][k], ref.P[k])
        np.testing.assert_array_equal(r1[1][k], ref.P[k])
        np.testing.assert_array_equal(r1[2][k], ref.NT[k])
    np.testing.assert_array_equal(r1[3], np.zeros(12, np.float32))     # ... and rank 0's optimiser state
    assert r0[4] == r1[4] == 2
    assert r0[5] == r1[5] == ["D_dem", "D_y2", "G"]                    # derived state rebuilt after the broadcast
    # one collective per update: gradient + loss pieces in one message, summed
    assert r0[9] == r1[9] == 2                                        # (attach's broadcasts are not all-reduces)
    tot = np.float32(_sums_for(0)) + np.float32(_sums_for(1))
    np.testing.assert_allclose(r0[6], tot, rtol=1e-6)
    np.testing.assert_array_equal(r0[6], r1[6])                        # identical pieces -> identical scalars / arg-min
    np.testing.assert_allclose(combine_generator_sums(r0[6]), combine_generator_sums(tot), rtol=1e-12)
    np.testing.assert_allclose(combine_critic_sums(r0[7]), combine_critic_sums([tot[0], tot[1], 3.0, 8.0]), rtol=1e-6)
    # the update equals a single process on the GLOBAL batch (mean over both shards = 1.5), replicas stay identical
    one = NumpyEngine(seed=0)
    one.update("G", np.concatenate([np.full((4, 2), 1.0), np.full((4, 2), 2.0)]).astype(np.float32), np.zeros(8))
    np.testing.assert_allclose(r0[8]["G"], one.P["G"], rtol=1e-6)
    for k in NumpyEngine.NETS:
        np.testing.assert_array_equal(r0[8][k], r1[8][k])


def test_m3_m4_are_not_rank_averages():
    a, b = _sums_for(0), _sums_for(1)
    tot = [x + y for x, y in zip(a, b)]
    glob = combine_generator_sums(tot)
    avg = [(x + y) / 2 for x, y in zip(combine_generator_sums(a), combine_generator_sums(b))]
    assert abs(glob[4] - avg[4]) > 1e-9      # M3: square of a global count difference
    assert abs(glob[1] - avg[1]) < 1e-12     # plain batch means do average


# ---- rank-sharded schedule: the union of the ranks' feeds is the single-process feed ----
class FeedRecorder:
    def __init__(self):
        self.feeds = []

    def _rec(self, kind, arrays):
        self.feeds.append((kind, [np.array(a, np.float64) for a in arrays]))

    def netD_y2_train(self, inp):
        self._rec("y2", inp)
        return [0.0, 0.0]

    def netD_dem_train(self, inp):
        self._rec("dem", inp)
        return [0.0, 0.0]

    def netG_no_update_many(self, inp):
        x, y2, zs = inp
        self._rec("evals", [x, y2, zs])
        return [[float(k == 3) * -1.0, 0, 0, 0, 0, 0] for k in range(len(zs))]      # noise 3 always wins

    def netG_train(self, inp):
        self._rec("train", inp)
        return [0.0] * 6


def _epoch_feeds(rank, world, bs):
    n = 6 * bs * world
    x = np.arange(n, dtype=np.float32).reshape(n, 1, 1, 1) * np.ones((1, 2, 2, 1), np.float32)
    y = x + 0.5
    rec, st = FeedRecorder(), ScheduleState()
    st.gen_iterations = 30
    train_epoch(rec, x, y, batchSize=bs, Diters=2, k_noise=4, state=st, rng=np.random.RandomState(7), rank=rank,
                world=world, fused=False)
    return rec.feeds, st


def test_rank_shards_union_equals_single_process_batches():
    world, bs = 2, 3
    single, st1 = _epoch_feeds(0, 1, bs * world)
    parts = [_epoch_feeds(r, world, bs) for r in range(world)]
    assert all(p[1].gen_iterations == st1.gen_iterations and p[1].crit_iterations == st1.crit_iterations
               for p in parts)
    assert [k for k, _ in single] == [k for k, _ in parts[0][0]] == [k for k, _ in parts[1][0]]
    for idx, (kind, arrays) in enumerate(single):
        for a, pieces in zip(arrays, zip(*[p[0][idx][1] for p in parts])):
            axis = 1 if (kind == "evals" and a.ndim == 4 and a.shape[-1] == 1 and a.shape[-2] == 32) else 0
            np.testing.assert_array_equal(a, np.concatenate(pieces, axis=axis))


class FusedRecorder(FeedRecorder):
    """Offers gen_iteration: replays it as the closure sequence so both schedules can be compared feed by feed."""

    def gen_iteration(self, y2_loop, dem_loop, gen, batch_stride=None):
        bs = gen[0].shape[0]
        stride = batch_stride or bs
        outs = []
        for kind, (x, y2, z, ep, n) in (("y2", y2_loop), ("dem", dem_loop)):
            o = []
            for j in range(n):
                self._rec(kind, [y2[j * stride:j * stride + bs], x[j * stride:j * stride + bs], z[j], ep[j]])
                o.append([0.0, 0.0])
            outs.append(o)
        ev = self.netG_no_update_many(list(gen))
        best = int(np.argmin([e[0] for e in ev]))
        tr = self.netG_train([gen[0], gen[1], gen[2][best]])
        return outs[0], outs[1], ev, tr, best


def test_fused_schedule_feeds_what_the_closure_schedule_feeds():
    for world, rank in ((1, 0), (2, 1)):
        bs = 3
        n = 7 * bs * world                                  # 7 global batches: the last critic loops are truncated
        x = np.arange(n, dtype=np.float32).reshape(n, 1, 1, 1) * np.ones((1, 2, 2, 1), np.float32)
        logs = []
        recs = []
        for fused, cls in ((False, FeedRecorder), (True, FusedRecorder)):
            rec, st, log = cls(), ScheduleState(), []
            st.gen_iterations = 30
            train_epoch(rec, x, x + 0.5, batchSize=bs, Diters=3, k_noise=4, state=st, rng=np.random.RandomState(3),
                        on_gen_iteration=log.append, rank=rank, world=world, fused=fused)
            recs.append(rec.feeds)
            logs.append([(d["i"], d["ii"], d["best_noise"], d["Diters"]) for d in log])
        assert logs[0] == logs[1] and len(logs[0]) == 3
        assert [k for k, _ in recs[0]] == [k for k, _ in recs[1]]
        for (k0, a0), (k1, a1) in zip(*recs):
            for u, v in zip(a0, a1):
                np.testing.assert_array_equal(u, v)


def test_bench_launches_its_own_ranks():
    """`python bench.py --gpus 2` with no launcher around it must start two ranks itself;
    --dry-run swaps the GPU engine for a host stand-in and RCCL for gloo."""
    env = dict(os.environ)
    for k in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT"):
        env.pop(k, None)
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--dry-run", "--steps", "3",
                        "--warmup", "1"], capture_output=True, text=True, timeout=300, env=env)
    assert r.returncode == 0, r.stderr[-2000:]
    lines = [ln for ln in r.stdout.splitlines() if ln.startswith("{")]
    assert len(lines) == 1                                   # rank 0's line only
    d = json.loads(lines[0])
    assert d["n_gpus"] == 2 and d["config"]["global_batch"] == 64 and d["config"]["parallelism"] == "dp2"
    assert d["dry_run"] is True and d["steps"] == 3
    # a launcher/--gpus mismatch is an error, not a silently single-GPU number
    env2 = dict(env, WORLD_SIZE="1", RANK="0", LOCAL_RANK="0")
    r2 = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "4", "--dry-run"],
                        capture_output=True, text=True, timeout=120, env=env2)
    assert r2.returncode != 0 and "WORLD_SIZE" in r2.stderr
