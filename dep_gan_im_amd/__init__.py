"""dep_gan_im_amd -- MI355X-native DEP-GAN two-critic WGAN-GP train step.

(The task names the package `dep-gan-im_amd`; hyphens are not importable, so the
directory is `dep_gan_im_amd`.)  Hand-written HIP kernels for gfx950 behind the
C ABI of include/depgan.h; this package is the Python host side that keeps the
reference's Keras-style call surface.
"""
import os as _os

# the host driver of the MI355X pool only supports dmabuf IPC (RCCL, sharing device tensors across processes); it has to
# be in the environment before the first HIP call of the process, so it is set when the package is imported
_os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")

from ._lib import DepganError, load  # noqa: F401,E402
from .models import Dis_C2D_FCN1, Gen_UNet2D  # noqa: F401
from .trainers import Trainers, build_trainers  # noqa: F401
from .engine import Engine  # noqa: F401
from . import evaluate  # noqa: F401,E402
from . import data, nifti  # noqa: F401,E402
