from .depthwise_student import DepthwiseStudent  # noqa: F401
from .transform_blocks import DepthwiseSeparableBlock  # noqa: F401
