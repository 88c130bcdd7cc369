"""Optimizers resolvable by name from configs (`config.init_obj('optimizer', utils.optim, params)`): every torch.optim
class plus the reference's RAdam (the reference's utils/optim/__init__.py star-imports torch.optim the same way)."""
import importlib

from torch.optim import *  # noqa: F401,F403

from .radam import RAdam  # noqa: F401

# torch.optim's star import also binds the name `lr_scheduler`; rebind it to this package's module
lr_scheduler = importlib.import_module(".lr_scheduler", __name__)
