from .meta_arch import META_ARCH_REGISTRY, build_model  # noqa: F401
from .depth_net import DEPTH_NET_REGISTRY, build_depth_net  # noqa: F401
from .pose_net import POSE_NET_REGISTRY, build_pose_net  # noqa: F401
