from .build import DEPTH_NET_REGISTRY, build_depth_net  # noqa: F401
from .DepthResNet import DepthResNet  # noqa: F401
from .PackNet01 import PackNet01  # noqa: F401
