"""MI355X-native KD train step for DeepLabV3+/WRN-38 students with cheap (depthwise-separable) convs.

Host-side mirror of the reference's plug points (losses.*, models.students.DepthwiseStudent,
trainer.LayerwiseTrainer, utils.optim.RAdam ...) over the HIP kernels of include/kdcc.h.
Import as `kdcc_amd` (see kdcc_amd.py at the repo root).
"""
__version__ = "0.1.0"

from . import _lib  # noqa: F401

from . import losses, models, trainer, utils  # noqa: E402,F401
from .parse_config import ConfigParser  # noqa: E402,F401
