from .depthwise_student import DepthwiseStudent  # noqa: F401
from .transform_blocks import DepthwiseSeparableBlock  # noqa: F401
from .taylor_prune_student import GateLayer, TaylorPruneStudent  # noqa: F401
