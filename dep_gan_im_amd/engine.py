"""Engine: one libdepgan context (generator + two critics + optimiser state) on one GPU.

PyTorch-ROCm is used for device buffers of the *inputs* and for the stream
handle only; every FLOP runs in the hand-written HIP kernels behind the C ABI
(include/depgan.h).  Mirrors the slice of the Keras API the reference touches
(GT:513-598): see models.py / trainers.py for the user-facing names.
"""
from __future__ import annotations

import ctypes as C
import os
from collections import OrderedDict

import numpy as np

from . import _lib
from ._lib import (ARENA_ADAM_M, ARENA_ADAM_V, ARENA_GRADS, ARENA_NONTRAINABLE, ARENA_PARAMS, D2H, H2D, NET_D_DEM,
                   NET_D_Y2, NET_G, Config, check, load)

NET_IDS = {"G": NET_G, "D_y2": NET_D_Y2, "D_dem": NET_D_DEM}


def _torch():
    import torch
    return torch


class Engine:
    def __init__(self, batch, height=256, width=256, nicg=1, first_fm=32, im_thresh=0.5, delta=10.0, lrD=1e-4,
                 lrG=1e-4, beta1=0.0, beta2=0.9, adam_eps=1e-7, device=None, nc_out=1, bf16_weights=False,
                 bf16_mfma=False, f32_split=0):
        torch = _torch()
        if not torch.cuda.is_available():
            raise _lib.DepganError("dep_gan_im_amd needs a ROCm GPU (MI355X): torch.cuda.is_available() is False")
        self.lib = load()
        self.device = torch.device(device if device is not None else "cuda:%d" % torch.cuda.current_device())
        torch.cuda.set_device(self.device)
        if not f32_split and not (bf16_weights or bf16_mfma) and nc_out == 1:
            # DEPGAN_F32_SPLIT=6 (or 3) turns the opt-in split-product convolutions on for contexts that do not ask for
            # anything else: lets the whole parity suite / bench run on them unchanged
            f32_split = int(os.environ.get("DEPGAN_F32_SPLIT", "0") or 0)
        self.f32_split = int(f32_split)
        self.cfg = Config(batch=batch, height=height, width=width, nicg=nicg, first_fm=first_fm, im_thresh=im_thresh,
                          delta=delta, lrD=lrD, lrG=lrG, beta1=beta1, beta2=beta2, adam_eps=adam_eps, nc_out=nc_out,
                          bf16_weights=1 if (bf16_weights or bf16_mfma) else 0, bf16_mfma=1 if bf16_mfma else 0,
                          f32_split=int(f32_split))
        self.batch, self.height, self.width, self.nicg, self.nc_out = batch, height, width, nicg, nc_out
        h = C.c_void_p()
        check(self.lib.depgan_create(C.byref(self.cfg), C.byref(h)), "depgan_create")
        self.h = h
        self._use_current_stream()
        self._tables = {}
        self._ar_cb = None       # keeps the ctypes callback of set_allreduce alive
        self.world = 1

    # ---- plumbing ----
    def _use_current_stream(self):
        torch = _torch()
        s = torch.cuda.current_stream(self.device).cuda_stream
        check(self.lib.depgan_set_stream(self.h, C.c_void_p(s)), "depgan_set_stream")

    def close(self):
        if getattr(self, "h", None):
            self.lib.depgan_destroy(self.h)
            self.h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def _dev(self, a, shape=None):
        """float32 contiguous CUDA tensor for a numpy array / tensor (cast like Keras' feed)."""
        torch = _torch()
        if isinstance(a, torch.Tensor):
            t = a.to(device=self.device, dtype=torch.float32).contiguous()
        else:
            t = torch.from_numpy(np.ascontiguousarray(np.asarray(a), dtype=np.float32)).to(self.device)
        if shape is not None and tuple(t.shape) != tuple(shape):
            raise ValueError("expected input of shape %s, got %s" % (tuple(shape), tuple(t.shape)))
        return t

    @staticmethod
    def _p(t):
        return C.c_void_p(t.data_ptr())

    # ---- parameter tables ----
    def param_table(self, net):
        nid = NET_IDS[net]
        if nid in self._tables:
            return self._tables[nid]
        n = self.lib.depgan_param_count(self.h, nid)
        out = []
        name = C.create_string_buffer(128)
        shape = (C.c_int * 4)()
        ndim, off, tr = C.c_int(), C.c_long(), C.c_int()
        for i in range(n):
            check(self.lib.depgan_param_info(self.h, nid, i, name, 128, shape, C.byref(ndim), C.byref(off),
                                             C.byref(tr)), "depgan_param_info")
            out.append((name.value.decode(), tuple(shape[:ndim.value]), off.value, bool(tr.value)))
        self._tables[nid] = out
        return out

    def _arena_np(self, net, arena):
        nid = NET_IDS[net]
        n = self.lib.depgan_arena_floats(self.h, nid, arena)
        buf = np.empty(max(n, 1), np.float32)
        ptr = self.lib.depgan_arena_ptr(self.h, nid, arena)
        _torch().cuda.synchronize(self.device)
        if n:
            _lib.memcpy(buf.ctypes.data, ptr, n * 4, D2H)
        return buf

    def _read(self, net, trainable_arena):
        tr = self._arena_np(net, trainable_arena)
        out = OrderedDict()
        nt = None
        for name, shape, off, trainable in self.param_table(net):
            size = int(np.prod(shape))
            if trainable:
                out[name] = tr[off:off + size].reshape(shape).copy()
            elif trainable_arena == ARENA_PARAMS:
                if nt is None:
                    nt = self._arena_np(net, ARENA_NONTRAINABLE)
                out[name] = nt[off:off + size].reshape(shape).copy()
        return out

    def get_weights(self, net):
        """OrderedDict name -> ndarray (Keras layouts), incl. BN moving statistics."""
        return self._read(net, ARENA_PARAMS)

    def get_grads(self, net):
        return self._read(net, ARENA_GRADS)

    def get_adam_state(self, net):
        return self._read(net, ARENA_ADAM_M), self._read(net, ARENA_ADAM_V)

    def set_weights(self, net, weights):
        nid = NET_IDS[net]
        tr = self._arena_np(net, ARENA_PARAMS)
        nt = self._arena_np(net, ARENA_NONTRAINABLE)
        for name, shape, off, trainable in self.param_table(net):
            if name not in weights:
                continue
            v = np.asarray(weights[name], np.float32)
            if tuple(v.shape) != tuple(shape):
                raise ValueError("weight %s: expected shape %s, got %s" % (name, shape, v.shape))
            (tr if trainable else nt)[off:off + v.size] = v.reshape(-1)
        unknown = set(weights) - {p[0] for p in self.param_table(net)}
        if unknown:
            raise ValueError("unknown weight names for %s: %s" % (net, sorted(unknown)[:5]))
        for arena, buf in ((ARENA_PARAMS, tr), (ARENA_NONTRAINABLE, nt)):
            n = self.lib.depgan_arena_floats(self.h, nid, arena)
            if n:
                _lib.memcpy(self.lib.depgan_arena_ptr(self.h, nid, arena), buf.ctypes.data, n * 4, H2D)
        self._use_current_stream()
        check(self.lib.depgan_weights_changed(self.h, nid), "depgan_weights_changed")

    def arena(self, net, arena):
        """(device pointer, number of floats) of one of a network's flat arenas (weights broadcast, checkpoints)."""
        nid = NET_IDS[net]
        return self.lib.depgan_arena_ptr(self.h, nid, arena), self.lib.depgan_arena_floats(self.h, nid, arena)

    def weights_changed(self, net):
        """Call after writing into a PARAMS / NONTRAINABLE arena from outside: rebuilds the derived state."""
        self._use_current_stream()
        check(self.lib.depgan_weights_changed(self.h, NET_IDS[net]), "depgan_weights_changed")

    def set_arena(self, net, arena, values):
        """Overwrites one flat arena from a float32 array of exactly its length (optimiser state on resume)."""
        ptr, n = self.arena(net, arena)
        v = np.ascontiguousarray(values, np.float32).reshape(-1)
        if v.size != n:
            raise ValueError("arena of %s has %d floats, got %d" % (net, n, v.size))
        _torch().cuda.synchronize(self.device)
        if n:
            _lib.memcpy(ptr, v.ctypes.data, n * 4, H2D)

    def get_arena(self, net, arena):
        return self._arena_np(net, arena)[:self.lib.depgan_arena_floats(self.h, NET_IDS[net], arena)].copy()

    def adam_step(self, net, value=None):
        """Adam `iterations` of a network's optimiser; with `value` sets it (resume)."""
        if value is not None:
            check(self.lib.depgan_set_adam_step(self.h, NET_IDS[net], int(value)), "depgan_set_adam_step")
        return int(self.lib.depgan_get_adam_step(self.h, NET_IDS[net]))

    def set_allreduce(self, fn, world):
        """Registers the data-parallel hook: fn(dev_ptr:int, n:int, stream:int) must ENQUEUE an in-place summing
        all-reduce of n device floats on the stream (dist.DataParallel.attach does this).  fn=None removes it."""
        if fn is None:
            self._ar_cb, self.world = None, 1
            check(self.lib.depgan_set_allreduce(self.h, _lib.ALLREDUCE_FN(), None, 1), "depgan_set_allreduce")
            return
        self._ar_err = None

        def _cb(user, ptr, n, stream):
            try:
                fn(int(ptr), int(n), int(stream or 0))
                return 0
            except BaseException as e:   # never let an exception unwind through the C frames
                self._ar_err = e
                return 1

        self._ar_cb = _lib.ALLREDUCE_FN(_cb)
        self.world = int(world)
        check(self.lib.depgan_set_allreduce(self.h, self._ar_cb, None, int(world)), "depgan_set_allreduce")

    # ---- direct RCCL binding (include/depgan.h): the library calls ncclAllReduce itself ----
    def rccl_unique_id(self):
        """The 128-byte ncclUniqueId (rank 0 creates it; the host hands it to every rank)."""
        buf = C.create_string_buffer(_lib.RCCL_ID_BYTES)
        check(self.lib.depgan_rccl_unique_id(buf), "depgan_rccl_unique_id")
        return buf.raw

    def rccl_init(self, uid, rank, world):
        """Collective: ncclCommInitRank on this engine's device.  Every update then all-reduces inside the library."""
        if len(uid) != _lib.RCCL_ID_BYTES:
            raise ValueError("an ncclUniqueId is %d bytes" % _lib.RCCL_ID_BYTES)
        _torch().cuda.set_device(self.device)
        self._ar_cb = None
        check(self.lib.depgan_rccl_init(self.h, C.c_char_p(uid), int(rank), int(world)), "depgan_rccl_init")
        self.world = int(world)

    def rccl_broadcast(self, ptr, n, root=0):
        self._use_current_stream()
        check(self.lib.depgan_rccl_broadcast(self.h, C.c_void_p(ptr), int(n), int(root)), "depgan_rccl_broadcast")

    def rccl_info(self):
        """(nranks, rank) as RCCL reports them, collectives issued by this engine."""
        n, r, k = C.c_int(), C.c_int(), C.c_long()
        check(self.lib.depgan_rccl_info(self.h, C.byref(n), C.byref(r), C.byref(k)), "depgan_rccl_info")
        return n.value, r.value, k.value

    def rccl_shutdown(self):
        check(self.lib.depgan_rccl_shutdown(self.h), "depgan_rccl_shutdown")
        self.world = 1 if self._ar_cb is None else self.world

    def _check(self, rc, what):
        err = getattr(self, "_ar_err", None)
        if err is not None:
            self._ar_err = None
            raise err
        check(rc, what)

    def grad_arena(self, net):
        """(device pointer, number of floats) of a network's flat gradient arena (for the RCCL all-reduce)."""
        nid = NET_IDS[net]
        return self.lib.depgan_arena_ptr(self.h, nid, ARENA_GRADS), self.lib.depgan_arena_floats(self.h, nid, ARENA_GRADS)

    # ---- forward ----
    def g_forward(self, x, z):
        torch = _torch()
        x = self._dev(x)
        z = self._dev(z).reshape(x.shape[0], -1)
        if x.dim() != 4 or tuple(x.shape[1:]) != (self.height, self.width, self.nicg):
            raise ValueError("generator input must be (N,%d,%d,%d), got %s" % (self.height, self.width, self.nicg,
                                                                                tuple(x.shape)))
        if z.shape[1] != 32:
            raise ValueError("noise input must be (N,32,1)")
        n = x.shape[0]
        out = torch.empty((n, self.height, self.width, self.nc_out), dtype=torch.float32, device=self.device)
        self._use_current_stream()
        for i in range(0, n, self.batch):
            m = min(self.batch, n - i)
            check(self.lib.depgan_g_forward(self.h, self._p(x[i:i + m]), self._p(z[i:i + m]), self._p(out[i:i + m]), m),
                  "depgan_g_forward")
        return out

    def d_forward(self, net, img):
        torch = _torch()
        img = self._dev(img)
        if img.dim() != 4 or tuple(img.shape[1:]) != (self.height, self.width, 1):
            raise ValueError("critic input must be (N,%d,%d,1), got %s" % (self.height, self.width, tuple(img.shape)))
        n = img.shape[0]
        out = torch.empty((n, 1), dtype=torch.float32, device=self.device)
        self._use_current_stream()
        cap = 3 * self.batch
        for i in range(0, n, cap):
            m = min(cap, n - i)
            check(self.lib.depgan_d_forward(self.h, NET_IDS[net], self._p(img[i:i + m]), self._p(out[i:i + m]), m),
                  "depgan_d_forward")
        return out

    # ---- closures ----
    def _batch_inputs(self, x, y2, z, ep=None):
        B = self.batch
        x = self._dev(x, (B, self.height, self.width, self.nicg))
        y2 = self._dev(y2, (B, self.height, self.width, 1))
        z = self._dev(z).reshape(-1)
        if z.numel() != B * 32:
            raise ValueError("noise must be (%d,32,1)" % B)
        if ep is not None:
            ep = self._dev(ep).reshape(-1)
            if ep.numel() != B:
                raise ValueError("ep must be (%d,1,1,1)" % B)
        return x, y2, z, ep

    def critic(self, which, y2, x, z, ep, update=True):
        x, y2, z, ep = self._batch_inputs(x, y2, z, ep)
        out = (C.c_float * 2)()
        self._use_current_stream()
        fn = self.lib.depgan_critic_step if update else self.lib.depgan_critic_grads
        self._check(fn(self.h, NET_IDS[which], self._p(y2), self._p(x), self._p(z), self._p(ep), out), "critic step")
        return [float(out[0]), float(out[1])]

    def generator(self, x, y2, z, mode="eval"):
        x, y2, z, _ = self._batch_inputs(x, y2, z)
        out = (C.c_float * 6)()
        self._use_current_stream()
        fn = {"eval": self.lib.depgan_g_eval, "grads": self.lib.depgan_g_grads, "step": self.lib.depgan_g_step}[mode]
        self._check(fn(self.h, self._p(x), self._p(y2), self._p(z), out), "generator " + mode)
        return [float(v) for v in out]

    def generator_eval_multi(self, x, y2, zs):
        """k forward-only loss evaluations on one batch with k noises (GT:868-877), one host sync.
        zs: (k, B, 32, 1) array / tensor or a list of k (B,32,1) noises.  Returns (k x 6 outputs, k x 8 sums)."""
        torch = _torch()
        B = self.batch
        x = self._dev(x, (B, self.height, self.width, self.nicg))
        y2 = self._dev(y2, (B, self.height, self.width, 1))
        if isinstance(zs, (list, tuple)):
            zs = torch.stack([self._dev(z).reshape(B, 32) for z in zs])
        else:
            zs = self._dev(zs)
        k = int(zs.shape[0])
        zs = zs.reshape(k, -1).contiguous()
        if zs.shape[1] != B * 32:
            raise ValueError("noises must be (k,%d,32,1)" % B)
        out, sums = (C.c_float * (6 * k))(), (C.c_float * (8 * k))()
        self._use_current_stream()
        self._check(self.lib.depgan_g_eval_multi(self.h, self._p(x), self._p(y2), self._p(zs), k, out, sums),
                    "depgan_g_eval_multi")
        return ([[float(out[6 * i + j]) for j in range(6)] for i in range(k)],
                [[float(sums[8 * i + j]) for j in range(8)] for i in range(k)])

    def gen_iteration(self, y2_loop, dem_loop, gen, batch_stride=None):
        """One generator iteration of the reference schedule (GT:791-878) with ONE host synchronisation
        (depgan_gen_iteration).

        y2_loop / dem_loop: (x, y2, z, ep, n) -- n consecutive batches: x (n*stride.., H, W, nicg) and y2 device
        tensors whose batch j starts at sample j*batch_stride, z (n, B, 32[,1]), ep (n, B[,1,1,1]); n may be 0.
        gen: (x, y2, zs) -- one batch and its k noises (k, B, 32[,1]).
        Returns (critic_y2 outs n x 2, critic_dem outs n x 2, eval outs k x 6, train out 6, best index)."""
        torch = _torch()
        B = self.batch
        stride = B if batch_stride is None else int(batch_stride)

        def loop(t):
            x, y2, z, ep, n = t
            n = int(n)
            if n == 0:
                return None, None, None, None, 0
            x, y2 = self._dev(x), self._dev(y2)
            need = (n - 1) * stride + B
            if x.shape[0] < need or y2.shape[0] < need or tuple(x.shape[1:]) != (self.height, self.width, self.nicg) \
                    or tuple(y2.shape[1:]) != (self.height, self.width, 1):
                raise ValueError("critic loop: need %d samples of (%d,%d,%d) / (%d,%d,1), got %s / %s"
                                 % (need, self.height, self.width, self.nicg, self.height, self.width,
                                    tuple(x.shape), tuple(y2.shape)))
            z, ep = self._dev(z).reshape(-1), self._dev(ep).reshape(-1)
            if z.numel() != n * B * 32 or ep.numel() != n * B:
                raise ValueError("critic loop: noise must be (%d,%d,32,1) and ep (%d,%d,1,1,1)" % (n, B, n, B))
            return x, y2, z, ep, n

        xa, ya, za, ea, na = loop(y2_loop)
        xb, yb, zb, eb, nb = loop(dem_loop)
        xg, yg, zs = gen
        xg = self._dev(xg, (B, self.height, self.width, self.nicg))
        yg = self._dev(yg, (B, self.height, self.width, 1))
        zs = self._dev(zs)
        k = int(zs.shape[0])
        zs = zs.reshape(k, -1).contiguous()
        if zs.shape[1] != B * 32:
            raise ValueError("noises must be (k,%d,32,1)" % B)
        if na + nb > _lib.MAX_CRITIC_STEPS or not 1 <= k <= _lib.MAX_MULTI:
            raise ValueError("gen_iteration: at most %d critic updates and %d noises" % (_lib.MAX_CRITIC_STEPS,
                                                                                       _lib.MAX_MULTI))
        nout = 2 * (na + nb) + 6 * k + 6
        out, best = (C.c_float * nout)(), C.c_int(-1)
        p = lambda t: self._p(t) if t is not None else C.c_void_p(0)   # noqa: E731
        self._use_current_stream()
        self._check(self.lib.depgan_gen_iteration(self.h, p(xa), p(ya), p(za), p(ea), na, p(xb), p(yb), p(zb), p(eb),
                                                  nb, stride, self._p(xg), self._p(yg), self._p(zs), k, out,
                                                  C.byref(best)), "depgan_gen_iteration")
        v = [float(t) for t in out]
        o = 0
        cy = [v[o + 2 * j:o + 2 * j + 2] for j in range(na)]
        o += 2 * na
        cd = [v[o + 2 * j:o + 2 * j + 2] for j in range(nb)]
        o += 2 * nb
        ev = [v[o + 6 * i:o + 6 * i + 6] for i in range(k)]
        o += 6 * k
        return cy, cd, ev, v[o:o + 6], int(best.value)

    # ---- DEP-UResNet supervised path (nc_out = 4) ----
    def uresnet(self, x, z, labels, mode="step", drop_seed=0):
        """mode 'step' = train_on_batch (UT:602-606), 'grads' = gradients only, 'eval' = phase-0 loss.
        The leading dimension may be shorter than the engine batch (Keras' last batch of an epoch)."""
        x = self._dev(x)
        n = int(x.shape[0])
        if n < 1 or n > self.batch or tuple(x.shape[1:]) != (self.height, self.width, self.nicg):
            raise ValueError("images must be (n<=%d,%d,%d,%d), got %s" % (self.batch, self.height, self.width,
                                                                          self.nicg, tuple(x.shape)))
        labels = self._dev(labels, (n, self.height, self.width, self.nc_out))
        z = self._dev(z).reshape(-1)
        if z.numel() != n * 32:
            raise ValueError("noise must be (%d,32,1)" % n)
        loss = C.c_float()
        self._use_current_stream()
        if mode == "eval":
            check(self.lib.depgan_uresnet_eval(self.h, self._p(x), self._p(z), self._p(labels), n, C.byref(loss)),
                  "depgan_uresnet_eval")
        else:
            fn = {"step": self.lib.depgan_uresnet_step, "grads": self.lib.depgan_uresnet_grads}[mode]
            check(fn(self.h, self._p(x), self._p(z), self._p(labels), n, C.c_uint(int(drop_seed) & 0xFFFFFFFF),
                     C.byref(loss)), "depgan_uresnet_" + mode)
        return float(loss.value)

    def apply_adam(self, net):
        self._use_current_stream()
        check(self.lib.depgan_apply_adam(self.h, NET_IDS[net]), "depgan_apply_adam")

    def last_sums(self):
        out = (C.c_float * 8)()
        check(self.lib.depgan_last_sums(self.h, out), "depgan_last_sums")
        return [float(v) for v in out]

    # ---- parity-test surface ----
    def debug_capture(self, on=True):
        """Keep a copy of the mixed pass's activations in every critic closure (depgan_debug_capture)."""
        check(self.lib.depgan_debug_capture(self.h, 1 if on else 0), "depgan_debug_capture")

    def debug_tensor(self, name):
        """An internal tensor of the last closure as a dense (N,H,W,C) float32 array (depgan_debug_tensor)."""
        shape = (C.c_int * 4)()
        check(self.lib.depgan_debug_tensor(self.h, name.encode(), None, 0, shape), "depgan_debug_tensor")
        out = np.empty(tuple(shape), np.float32)
        check(self.lib.depgan_debug_tensor(self.h, name.encode(), C.c_void_p(out.ctypes.data), out.size, shape),
              "depgan_debug_tensor")
        return out

    # ---- profiling ----
    def profile(self, on):
        check(self.lib.depgan_profile_enable(self.h, 1 if on else 0))

    def profile_reset(self):
        check(self.lib.depgan_profile_reset(self.h))

    def profile_dump(self, path):
        check(self.lib.depgan_profile_dump(self.h, path.encode()))

    def profile_read(self, klass):
        ms, n, fl = C.c_double(), C.c_long(), C.c_double()
        check(self.lib.depgan_profile_read(self.h, klass, C.byref(ms), C.byref(n), C.byref(fl)))
        return ms.value, n.value, fl.value

    def profile_read_bytes(self, klass):
        by = C.c_double()
        check(self.lib.depgan_profile_read_bytes(self.h, klass, C.byref(by)))
        return by.value
