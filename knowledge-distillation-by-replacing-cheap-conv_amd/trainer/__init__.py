from .classification_trainer import ClassificationTrainer  # noqa: F401
from .layerwise_trainer import LayerwiseTrainer  # noqa: F401
from .taylor_prune_trainer import TaylorPruneTrainer  # noqa: F401
