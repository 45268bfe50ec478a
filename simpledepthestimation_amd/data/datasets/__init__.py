from .kitti_v2 import KittiDepthV2  # noqa: F401
