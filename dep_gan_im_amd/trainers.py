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

from .engine import Engine


class Trainers:
    def __init__(self, engine, dist=None):
        self.engine = engine
        self.dist = dist  # optional dep_gan_im_amd.dist.DataParallel

    # ---- critics ----
    def _critic(self, which, inputs):
        if not isinstance(inputs, (list, tuple)) or len(inputs) != 4:
            raise ValueError("critic closure expects [real_2tp, real_1tp, noise, ep]")
        y2, x, z, ep = inputs
        if self.dist is None:
            return self.engine.critic(which, y2, x, z, ep, update=True)
        out = self.engine.critic(which, y2, x, z, ep, update=False)
        out = self.dist.reduce_critic(self.engine, which, out)
        self.engine.apply_adam(which)
        return out

    def netD_y2_train(self, inputs):
        return self._critic("D_y2", inputs)

    def netD_dem_train(self, inputs):
        return self._critic("D_dem", inputs)

    # ---- generator ----
    def netG_no_update(self, inputs):
        if not isinstance(inputs, (list, tuple)) or len(inputs) != 3:
            raise ValueError("generator closure expects [real_1tp, real_2tp, noise]")
        x, y2, z = inputs
        out = self.engine.generator(x, y2, z, "eval")
        if self.dist is not None:
            out = self.dist.reduce_generator(self.engine, out, grads=False)
        return out

    def netG_no_update_many(self, inputs):
        """[x, y2, [z_0 .. z_{k-1}]] -> k lists of the six netG_no_update scalars: the driver's best-of-k noise
        search (GT:868-877) as one enqueue with one host synchronisation instead of k closure calls."""
        if not isinstance(inputs, (list, tuple)) or len(inputs) != 3:
            raise ValueError("generator closure expects [real_1tp, real_2tp, noises]")
        x, y2, zs = inputs
        outs, sums = self.engine.generator_eval_multi(x, y2, zs)
        if self.dist is not None:
            outs = self.dist.reduce_generator_many(sums, getattr(self.engine, "device", None))
        return outs

    def netG_train(self, inputs):
        if not isinstance(inputs, (list, tuple)) or len(inputs) != 3:
            raise ValueError("generator closure expects [real_1tp, real_2tp, noise]")
        x, y2, z = inputs
        if self.dist is None:
            return self.engine.generator(x, y2, z, "step")
        out = self.engine.generator(x, y2, z, "grads")
        out = self.dist.reduce_generator(self.engine, out, grads=True)
        self.engine.apply_adam("G")
        return out

    # train_on_batch-style aliases
    def critic_y2_train_on_batch(self, real_2tp, real_1tp, noise, ep):
        return self.netD_y2_train([real_2tp, real_1tp, noise, ep])

    def critic_dem_train_on_batch(self, real_2tp, real_1tp, noise, ep):
        return self.netD_dem_train([real_2tp, real_1tp, noise, ep])

    def generator_train_on_batch(self, real_1tp, real_2tp, noise):
        return self.netG_train([real_1tp, real_2tp, noise])


def build_trainers(netG, netD_y2, netD_dem, batchSize=16, delta=10.0, lrD=1e-4, lrG=1e-4, IM_TRSH=0.5,
                   dist=None, device=None, weights_dtype="float32"):
    """Builds the loss graph of GT:523-598 for the three models and returns a
    Trainers object.  The models are bound to one engine: afterwards their
    predict()/get_weights()/save() see the trained weights.
    weights_dtype="bfloat16" (BASELINE config 4): kernels are rounded to bf16 before every use, fp32 accumulate,
    fp32 master weights and Adam state; get_weights() returns the fp32 masters."""
    if weights_dtype not in ("float32", "bfloat16"):
        raise ValueError("weights_dtype must be 'float32' or 'bfloat16'")
    H, W, nicg = netG.input_shape
    if tuple(netD_y2.input_shape) != (H, W, 1) or tuple(netD_dem.input_shape) != (H, W, 1):
        raise ValueError("critics must take (%d,%d,1) images" % (H, W))
    eng = Engine(batchSize, H, W, nicg, first_fm=netG.first_fm, im_thresh=IM_TRSH, delta=delta, lrD=lrD, lrG=lrG,
                 beta1=0.0, beta2=0.9, device=device, bf16_weights=(weights_dtype == "bfloat16"))
    netG._bind(eng, "G")
    netD_y2._bind(eng, "D_y2")
    netD_dem._bind(eng, "D_dem")
    return Trainers(eng, dist)
