"""ctypes binding of libdepgan.so (the C ABI declared in include/depgan.h).

There is no CPU fallback: if the HIP library is missing the import fails loudly.
"""
from __future__ import annotations

import ctypes as C
import os

HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get("DEPGAN_LIB") or os.path.join(HERE, "libdepgan.so")   # DEPGAN_LIB: A/B another build

EXPORTS = [
    "depgan_last_error", "depgan_create", "depgan_destroy", "depgan_set_stream", "depgan_param_count",
    "depgan_param_info", "depgan_arena_floats", "depgan_arena_ptr", "depgan_weights_changed", "depgan_g_forward",
    "depgan_d_forward", "depgan_critic_grads", "depgan_critic_step", "depgan_g_eval", "depgan_g_grads",
    "depgan_g_step", "depgan_apply_adam", "depgan_last_sums", "depgan_profile_enable", "depgan_profile_read",
    "depgan_profile_reset", "depgan_profile_dump", "depgan_op_conv2d", "depgan_op_conv2d_bwd_data", "depgan_op_conv2d_wgrad",
    "depgan_op_maxpool", "depgan_op_deconv2x2", "depgan_op_deconv2x2_wgrad", "depgan_op_conv2d_stamps", "depgan_uresnet_grads", "depgan_uresnet_step",
    "depgan_uresnet_eval", "depgan_profile_read_bytes", "depgan_g_eval_multi", "depgan_eval_accumulate", "depgan_eval_counts",
    "depgan_data_prep_scratch_floats", "depgan_data_prep_subject", "depgan_abi_version", "depgan_config_size",
    "depgan_set_allreduce", "depgan_get_adam_step", "depgan_set_adam_step", "depgan_gen_iteration", "depgan_eval_divide",
    "depgan_source_hash", "depgan_debug_capture", "depgan_debug_tensor", "depgan_rccl_unique_id", "depgan_rccl_init",
    "depgan_rccl_broadcast", "depgan_rccl_info", "depgan_rccl_shutdown", "depgan_op_conv2d_wgrad_bf16",
]

ABI_VERSION = 3          # DEPGAN_ABI_VERSION of the include/depgan.h this binding was written against
MAX_MULTI, MAX_CRITIC_STEPS = 32, 256
RCCL_ID_BYTES = 128


class Config(C.Structure):
    """depgan_config of include/depgan.h, field for field (tests/test_lib_cpu.py parses the header and compares)."""
    _fields_ = [("struct_size", C.c_int), ("batch", C.c_int), ("height", C.c_int), ("width", C.c_int),
                ("nicg", C.c_int), ("first_fm", C.c_int), ("im_thresh", C.c_float), ("delta", C.c_float),
                ("lrD", C.c_float), ("lrG", C.c_float), ("beta1", C.c_float), ("beta2", C.c_float),
                ("adam_eps", C.c_float), ("nc_out", C.c_int), ("bf16_weights", C.c_int), ("bf16_mfma", C.c_int),
                ("f32_split", C.c_int)]

    def __init__(self, **kw):
        super().__init__(**kw)
        self.struct_size = C.sizeof(Config)


# int fn(void* user, float* dev_ptr, long n, void* hip_stream): the all-reduce hook of depgan_set_allreduce
ALLREDUCE_FN = C.CFUNCTYPE(C.c_int, C.c_void_p, C.c_void_p, C.c_long, C.c_void_p)


NET_G, NET_D_Y2, NET_D_DEM = 0, 1, 2
ARENA_PARAMS, ARENA_NONTRAINABLE, ARENA_GRADS, ARENA_ADAM_M, ARENA_ADAM_V = range(5)

_lib = None


class DepganError(RuntimeError):
    pass


def load():
    """Load libdepgan.so; raises if it has not been built (python -m dep_gan_im_amd.build)."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise DepganError(
            "libdepgan.so not found at %s -- the HIP library is the product; build it with "
            "`python -m dep_gan_im_amd.build` (needs hipcc). There is no CPU fallback." % LIB_PATH)
    # PyTorch-ROCm ships its own HIP runtime.  It must be in the process BEFORE libdepgan.so is loaded, so that the
    # library binds to that runtime: loaded the other way round, two runtimes coexist and hipMalloc inside the
    # library reports "no ROCm-capable device is detected".
    import torch  # noqa: F401
    lib = C.CDLL(LIB_PATH)
    fp, vp, ip = C.POINTER(C.c_float), C.c_void_p, C.POINTER(C.c_int)
    lib.depgan_last_error.restype = C.c_char_p
    lib.depgan_abi_version.argtypes = []
    lib.depgan_config_size.argtypes = []
    lib.depgan_config_size.restype = C.c_size_t
    if lib.depgan_abi_version() != ABI_VERSION or lib.depgan_config_size() != C.sizeof(Config):
        raise DepganError("libdepgan.so at %s has ABI %d / depgan_config of %d bytes, this binding expects ABI %d / %d "
                          "bytes: rebuild with `python -m dep_gan_im_amd.build`"
                          % (LIB_PATH, lib.depgan_abi_version(), lib.depgan_config_size(), ABI_VERSION, C.sizeof(Config)))
    lib.depgan_source_hash.restype = C.c_char_p
    if os.path.isdir(os.path.join(HERE, "csrc")) and not os.environ.get("DEPGAN_LIB"):
        from .build import source_hash
        built, here = lib.depgan_source_hash().decode(), source_hash()
        if built != here:
            raise DepganError("libdepgan.so at %s was built from other sources (hash %s, the sources here hash to %s): "
                              "rebuild with `python -m dep_gan_im_amd.build`" % (LIB_PATH, built, here))
    lib.depgan_set_allreduce.argtypes = [vp, ALLREDUCE_FN, vp, C.c_int]
    lib.depgan_get_adam_step.argtypes = [vp, C.c_int]
    lib.depgan_get_adam_step.restype = C.c_long
    lib.depgan_set_adam_step.argtypes = [vp, C.c_int, C.c_long]
    lib.depgan_gen_iteration.argtypes = [vp, vp, vp, vp, vp, C.c_int, vp, vp, vp, vp, C.c_int, C.c_long, vp, vp, vp,
                                         C.c_int, fp, ip]
    lib.depgan_create.argtypes = [C.POINTER(Config), C.POINTER(vp)]
    lib.depgan_destroy.argtypes = [vp]
    lib.depgan_destroy.restype = None
    lib.depgan_set_stream.argtypes = [vp, vp]
    lib.depgan_param_count.argtypes = [vp, C.c_int]
    lib.depgan_param_info.argtypes = [vp, C.c_int, C.c_int, C.c_char_p, C.c_int, ip, ip, C.POINTER(C.c_long), ip]
    lib.depgan_arena_floats.argtypes = [vp, C.c_int, C.c_int]
    lib.depgan_arena_floats.restype = C.c_long
    lib.depgan_arena_ptr.argtypes = [vp, C.c_int, C.c_int]
    lib.depgan_arena_ptr.restype = vp
    lib.depgan_weights_changed.argtypes = [vp, C.c_int]
    lib.depgan_g_forward.argtypes = [vp, vp, vp, vp, C.c_int]
    lib.depgan_d_forward.argtypes = [vp, C.c_int, vp, vp, C.c_int]
    lib.depgan_critic_grads.argtypes = [vp, C.c_int, vp, vp, vp, vp, fp]
    lib.depgan_critic_step.argtypes = [vp, C.c_int, vp, vp, vp, vp, fp]
    lib.depgan_g_eval.argtypes = [vp, vp, vp, vp, fp]
    lib.depgan_g_grads.argtypes = [vp, vp, vp, vp, fp]
    lib.depgan_g_eval_multi.argtypes = [vp, vp, vp, vp, C.c_int, fp, fp]
    lib.depgan_g_step.argtypes = [vp, vp, vp, vp, fp]
    lib.depgan_apply_adam.argtypes = [vp, C.c_int]
    lib.depgan_last_sums.argtypes = [vp, fp]
    lib.depgan_profile_enable.argtypes = [vp, C.c_int]
    lib.depgan_profile_read.argtypes = [vp, C.c_int, C.POINTER(C.c_double), C.POINTER(C.c_long), C.POINTER(C.c_double)]
    lib.depgan_profile_read_bytes.argtypes = [vp, C.c_int, C.POINTER(C.c_double)]
    lib.depgan_profile_reset.argtypes = [vp]
    lib.depgan_profile_dump.argtypes = [vp, C.c_char_p]
    lib.depgan_op_conv2d.argtypes = [vp, vp, vp, vp] + [C.c_int] * 8 + [vp]
    lib.depgan_op_conv2d_bwd_data.argtypes = [vp, vp, vp] + [C.c_int] * 7 + [vp]
    lib.depgan_op_conv2d_wgrad.argtypes = [vp, vp, vp] + [C.c_int] * 6 + [vp]
    lib.depgan_op_conv2d_wgrad_bf16.argtypes = [vp, vp, vp] + [C.c_int] * 6 + [vp]
    lib.depgan_eval_accumulate.argtypes = [vp, vp, vp, C.c_long, vp]
    lib.depgan_eval_divide.argtypes = [vp, C.c_long, C.c_double, vp]
    lib.depgan_eval_counts.argtypes = [vp, C.c_int] + [vp] * 7 + [C.c_long, C.c_double, C.POINTER(C.c_longlong), vp]
    lib.depgan_data_prep_scratch_floats.argtypes = [C.c_int] * 3
    lib.depgan_data_prep_scratch_floats.restype = C.c_size_t
    lib.depgan_data_prep_subject.argtypes = [vp] * 7 + [C.c_int] * 4 + [vp] * 4
    lib.depgan_op_maxpool.argtypes = [vp, vp] + [C.c_int] * 4 + [vp]
    lib.depgan_op_deconv2x2.argtypes = [vp] * 6 + [C.c_int] * 6 + [vp]
    lib.depgan_op_deconv2x2_wgrad.argtypes = [vp] * 4 + [C.c_int] * 5 + [vp]
    lib.depgan_op_conv2d_stamps.argtypes = [vp, vp, vp] + [C.c_int] * 6 + [vp, C.c_int, vp]
    lib.depgan_uresnet_grads.argtypes = [vp, vp, vp, vp, C.c_int, C.c_uint, fp]
    lib.depgan_uresnet_step.argtypes = [vp, vp, vp, vp, C.c_int, C.c_uint, fp]
    lib.depgan_uresnet_eval.argtypes = [vp, vp, vp, vp, C.c_int, fp]
    lib.depgan_debug_capture.argtypes = [vp, C.c_int]
    lib.depgan_rccl_unique_id.argtypes = [vp]
    lib.depgan_rccl_init.argtypes = [vp, vp, C.c_int, C.c_int]
    lib.depgan_rccl_broadcast.argtypes = [vp, vp, C.c_long, C.c_int]
    lib.depgan_rccl_info.argtypes = [vp, ip, ip, C.POINTER(C.c_long)]
    lib.depgan_rccl_shutdown.argtypes = [vp]
    lib.depgan_debug_tensor.argtypes = [vp, C.c_char_p, vp, C.c_long, ip]
    _lib = lib
    return lib


def check(rc, what=""):
    if rc != 0:
        msg = load().depgan_last_error()
        raise DepganError("%s failed (status %d): %s" % (what or "libdepgan call", rc,
                                                         msg.decode() if msg else "?"))


_hip = None


def hip():
    """libamdhip64 for raw device copies (weights in/out of the C-side arenas)."""
    global _hip
    if _hip is None:
        last = None
        for name in ("libamdhip64.so", "/opt/rocm/lib/libamdhip64.so", "libamdhip64.so.7", "libamdhip64.so.6"):
            try:
                _hip = C.CDLL(name)
                break
            except OSError as e:  # pragma: no cover
                last = e
        if _hip is None:
            raise DepganError("cannot load libamdhip64: %s" % last)
        _hip.hipMemcpy.argtypes = [C.c_void_p, C.c_void_p, C.c_size_t, C.c_int]
        _hip.hipDeviceSynchronize.argtypes = []
    return _hip


H2D, D2H, D2D = 1, 2, 3


def memcpy(dst, src, nbytes, kind):
    rc = hip().hipMemcpy(C.c_void_p(dst), C.c_void_p(src), C.c_size_t(nbytes), C.c_int(kind))
    if rc != 0:
        raise DepganError("hipMemcpy failed with status %d" % rc)
