"""pose_vec2mat with the reference's name (detectron2/geometry/pose_utils.py:L130-137), computed by libsde_hip.so."""
from ..hip.photometric import pose_vec2mat  # noqa: F401
