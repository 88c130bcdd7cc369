"""LR schedulers resolvable by name from configs: torch's own plus `MyReduceLROnPlateau` (resettable plateau scheduler,
reference utils/optim/lr_scheduler/reduce_lr_on_plateau.py:6-178) and `MyOneCycleLR`."""
from torch.optim.lr_scheduler import *  # noqa: F401,F403
from torch.optim.lr_scheduler import OneCycleLR, ReduceLROnPlateau


class MyReduceLROnPlateau(ReduceLROnPlateau):
    """ReduceLROnPlateau that can be reset when new layers are unfrozen (LayerwiseTrainer.reset_scheduler)."""

    def __init__(self, optimizer, mode='min', factor=0.1, patience=10, verbose=False, threshold=1e-4,
                 threshold_mode='rel', cooldown=0, min_lr=0, eps=1e-8):
        super().__init__(optimizer, mode=mode, factor=factor, patience=patience, threshold=threshold,
                         threshold_mode=threshold_mode, cooldown=cooldown, min_lr=min_lr, eps=eps)
        self.verbose = verbose

    def reset(self):
        self._reset()


class MyOneCycleLR(OneCycleLR):
    pass
