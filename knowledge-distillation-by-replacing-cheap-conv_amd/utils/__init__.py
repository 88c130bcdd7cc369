from . import optim  # noqa: F401
from .weight_scheduler import WeightScheduler  # noqa: F401
