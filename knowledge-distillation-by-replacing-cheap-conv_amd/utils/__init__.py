from . import optim  # noqa: F401
from .util import (CityscapesMetricTracker, EarlyStopTracker, ImportanceFilterTracker, MetricTracker, ensure_dir, inf_loop, read_json,  # noqa: F401
                   write_json)
from .weight_scheduler import WeightScheduler  # noqa: F401
