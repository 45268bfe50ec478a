from .build import POSE_NET_REGISTRY, build_pose_net  # noqa: F401
from .PoseNet import PoseNet  # noqa: F401
