"""The four K.function closures of the reference (GT:549-598) over the HIP engine.

    netD_y2_train([y2, x, z, ep])  -> [loss_real, loss_fake]            GT:550-552, 809
    netD_dem_train([y2, x, z, ep]) -> [loss_real_dem, loss_fake_dem]    GT:569-571, 824
    netG_no_update([x, y2, z])     -> [loss, loss_fake, loss_fake_dem, M1, M3, M4]   GT:595-596, 873
    netG_train([x, y2, z])         -> same six, then Adam(theta_G)      GT:597-598, 878

Inputs may be NumPy arrays of any float dtype (the reference feeds float64
noise/ep, GT:807-808) or torch CUDA tensors already resident in HBM.
Outputs are Python floats computed with the pre-update weights, like
K.function.  `train_on_batch`-style aliases are provided because the
north-star text uses that name; the reference itself never calls it.
"""
from __future__ import annotations

import numpy as np

from .engine import Engine


class Trainers:
    def __init__(self, engine, dist=None):
        self.engine = engine
        self.dist = dist  # optional dep_gan_im_amd.dist.DataParallel, already attached to the engine

    # ---- critics ----
    def _critic(self, which, inputs):
        if not isinstance(inputs, (list, tuple)) or len(inputs) != 4:
            raise ValueError("critic closure expects [real_2tp, real_1tp, noise, ep]")
        y2, x, z, ep = inputs
        # with a DataParallel attached the library all-reduces gradient + loss pieces itself (one message, in stream
        # order) and the two scalars are the GLOBAL batch means
        return self.engine.critic(which, y2, x, z, ep, update=True)

    def netD_y2_train(self, inputs):
        return self._critic("D_y2", inputs)

    def netD_dem_train(self, inputs):
        return self._critic("D_dem", inputs)

    # ---- generator ----
    def netG_no_update(self, inputs):
        if not isinstance(inputs, (list, tuple)) or len(inputs) != 3:
            raise ValueError("generator closure expects [real_1tp, real_2tp, noise]")
        x, y2, z = inputs
        return self.engine.generator(x, y2, z, "eval")

    def netG_no_update_many(self, inputs):
        """[x, y2, [z_0 .. z_{k-1}]] -> k lists of the six netG_no_update scalars: the driver's best-of-k noise
        search (GT:868-877) as one enqueue with one host synchronisation instead of k closure calls."""
        if not isinstance(inputs, (list, tuple)) or len(inputs) != 3:
            raise ValueError("generator closure expects [real_1tp, real_2tp, noises]")
        x, y2, zs = inputs
        outs, _ = self.engine.generator_eval_multi(x, y2, zs)
        return outs

    def netG_train(self, inputs):
        if not isinstance(inputs, (list, tuple)) or len(inputs) != 3:
            raise ValueError("generator closure expects [real_1tp, real_2tp, noise]")
        x, y2, z = inputs
        return self.engine.generator(x, y2, z, "step")

    def gen_iteration(self, y2_loop, dem_loop, gen, batch_stride=None):
        """The whole generator iteration of the reference schedule (GT:791-829, 868-878) -- n critic-Y2 updates,
        n critic-DEM updates, the best-of-k noise search with its arg-min and the generator update -- as ONE enqueue
        with ONE host synchronisation.  Arguments and result as Engine.gen_iteration."""
        return self.engine.gen_iteration(y2_loop, dem_loop, gen, batch_stride)

    # ---- training state: the three networks, their optimisers and the schedule counters (GT:47-50, 892) ----
    def state_dict(self, schedule_state=None, extra=None):
        """Everything a resumed run needs to continue bit-identically: weights and BN moving statistics by their
        Keras names, Adam m / v arenas and `iterations` per optimiser, and the reference's module-level counters.
        extra: dict of arrays the DRIVER needs for the same guarantee (its RNG state, the data order, the epoch)."""
        from ._lib import ARENA_ADAM_M, ARENA_ADAM_V
        out = {}
        for net in ("G", "D_y2", "D_dem"):
            for k, v in self.engine.get_weights(net).items():
                out["%s/weights/%s" % (net, k)] = v
            out["%s/adam/m" % net] = self.engine.get_arena(net, ARENA_ADAM_M)
            out["%s/adam/v" % net] = self.engine.get_arena(net, ARENA_ADAM_V)
            out["%s/adam/iterations" % net] = np.asarray(self.engine.adam_step(net), np.int64)
        if schedule_state is not None:
            for k in ("gen_iterations", "crit_iterations", "crit_dem_iterations"):
                out["schedule/" + k] = np.asarray(getattr(schedule_state, k), np.int64)
            out["schedule/errG"] = np.asarray(schedule_state.errG, np.float64)
        for k, v in (extra or {}).items():
            out["extra/" + k] = np.asarray(v)
        return out

    def save_state(self, path, schedule_state=None, extra=None):
        np.savez(path, **self.state_dict(schedule_state, extra))

    @staticmethod
    def rng_to_arrays(rng):
        """np.random.RandomState -> dict of arrays for `extra` (and back: rng_from_arrays)."""
        name, keys, pos, has_gauss, cached = rng.get_state()
        return {"rng_keys": keys, "rng_pos": np.asarray([pos, has_gauss], np.int64), "rng_gauss": np.asarray(cached)}

    @staticmethod
    def rng_from_arrays(rng, extra):
        rng.set_state(("MT19937", np.asarray(extra["rng_keys"], np.uint32), int(extra["rng_pos"][0]),
                       int(extra["rng_pos"][1]), float(extra["rng_gauss"])))
        return rng

    def load_state(self, path_or_dict, schedule_state=None):
        from ._lib import ARENA_ADAM_M, ARENA_ADAM_V
        if isinstance(path_or_dict, dict):
            d = path_or_dict
        else:
            with np.load(path_or_dict) as f:
                d = {k: f[k] for k in f.files}
        for net in ("G", "D_y2", "D_dem"):
            pre = "%s/weights/" % net
            w = {k[len(pre):]: v for k, v in d.items() if k.startswith(pre)}
            names = [p[0] for p in self.engine.param_table(net)]
            missing = [n for n in names if n not in w]
            if missing:
                raise KeyError("training state lacks %d weights of %s, e.g. %s" % (len(missing), net, missing[:3]))
            self.engine.set_weights(net, w)
            self.engine.set_arena(net, ARENA_ADAM_M, d["%s/adam/m" % net])
            self.engine.set_arena(net, ARENA_ADAM_V, d["%s/adam/v" % net])
            self.engine.adam_step(net, int(d["%s/adam/iterations" % net]))
        if schedule_state is not None and "schedule/gen_iterations" in d:
            for k in ("gen_iterations", "crit_iterations", "crit_dem_iterations"):
                setattr(schedule_state, k, int(d["schedule/" + k]))
            schedule_state.errG = float(d["schedule/errG"])
        self.extra = {k[len("extra/"):]: v for k, v in d.items() if k.startswith("extra/")}
        return schedule_state

    # train_on_batch-style aliases
    def critic_y2_train_on_batch(self, real_2tp, real_1tp, noise, ep):
        return self.netD_y2_train([real_2tp, real_1tp, noise, ep])

    def critic_dem_train_on_batch(self, real_2tp, real_1tp, noise, ep):
        return self.netD_dem_train([real_2tp, real_1tp, noise, ep])

    def generator_train_on_batch(self, real_1tp, real_2tp, noise):
        return self.netG_train([real_1tp, real_2tp, noise])


def build_trainers(netG, netD_y2, netD_dem, batchSize=16, delta=10.0, lrD=1e-4, lrG=1e-4, IM_TRSH=0.5,
                   dist=None, device=None, weights_dtype="float32", activations_dtype="float32", f32_split=0):
    """Builds the loss graph of GT:523-598 for the three models and returns a
    Trainers object.  The models are bound to one engine: afterwards their
    predict()/get_weights()/save() see the trained weights.
    weights_dtype="bfloat16" (BASELINE config 4): kernels are rounded to bf16 before every use, fp32 accumulate,
    fp32 master weights and Adam state; get_weights() returns the fp32 masters.
    activations_dtype="bfloat16" (needs bf16 weights): the MFMA convolutions also round their activation operand to
    bf16 and run on the bf16 matrix pipe (v_mfma_f32_32x32x16_bf16), accumulating in fp32.
    f32_split=6 (or 3): opt-in -- fp32 operands of the MFMA convolutions split exactly into three (two) bf16 terms, the six
    (three) largest cross products on the bf16 matrix pipe, fp32 accumulation (include/depgan.h, depgan_config.f32_split).
    dist: a dep_gan_im_amd.dist.DataParallel -- the engine becomes one replica of a data-parallel job (rank 0's
    weights are broadcast, every update all-reduces its gradient arena)."""
    if weights_dtype not in ("float32", "bfloat16"):
        raise ValueError("weights_dtype must be 'float32' or 'bfloat16'")
    if activations_dtype not in ("float32", "bfloat16"):
        raise ValueError("activations_dtype must be 'float32' or 'bfloat16'")
    if activations_dtype == "bfloat16" and weights_dtype != "bfloat16":
        raise ValueError("activations_dtype='bfloat16' needs weights_dtype='bfloat16'")
    H, W, nicg = netG.input_shape
    if tuple(netD_y2.input_shape) != (H, W, 1) or tuple(netD_dem.input_shape) != (H, W, 1):
        raise ValueError("critics must take (%d,%d,1) images" % (H, W))
    eng = Engine(batchSize, H, W, nicg, first_fm=netG.first_fm, im_thresh=IM_TRSH, delta=delta, lrD=lrD, lrG=lrG,
                 beta1=0.0, beta2=0.9, device=device, bf16_weights=(weights_dtype == "bfloat16"),
                 bf16_mfma=(activations_dtype == "bfloat16"), f32_split=f32_split)
    netG._bind(eng, "G")
    netD_y2._bind(eng, "D_y2")
    netD_dem._bind(eng, "D_dem")
    if dist is not None:
        dist.attach(eng)
    return Trainers(eng, dist)
