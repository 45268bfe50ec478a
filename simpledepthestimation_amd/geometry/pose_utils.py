"""Pose helpers with the reference's names (detectron2/geometry/pose_utils.py).

  * torch side (L98-145): ``pose_vec2mat`` / ``euler2mat`` run on libsde_hip.so (sde_pose_vec2mat: R = Rx(a) Ry(b) Rz(c), the reference's
    ``xmat.bmm(ymat).bmm(zmat)``); ``invert_pose`` is three tiny device ops.
  * numpy side (L7-95), used by the KITTI reader for the OXTS odometry ground truth (datasets/kitti_v2.py:L128-132,L178-195): elementary
    rotations, the Mercator-projected translation of an OXTS packet and its roll / pitch / yaw rotation, homogeneous-transform helpers.
"""
import numpy as np
import torch

from ..hip.photometric import pose_vec2mat  # noqa: F401  (pose_utils.py:L130-137)

# field order of a KITTI raw ``oxts/data/*.txt`` line (dataformat.txt of the KITTI raw devkit); only the first six are used here
OXTS_FIELDS = ("lat", "lon", "alt", "roll", "pitch", "yaw", "vn", "ve", "vf", "vl", "vu", "ax", "ay", "az", "af", "al", "au", "wx", "wy", "wz", "wf",
               "wl", "wu", "pos_accuracy", "vel_accuracy", "navstat", "numsats", "posmode", "velmode", "orimode")
EARTH_RADIUS_M = 6378137.0


def rotx_np(t):
    c, s = np.cos(t), np.sin(t)
    return np.array([[1, 0, 0], [0, c, -s], [0, s, c]])


def roty_np(t):
    c, s = np.cos(t), np.sin(t)
    return np.array([[c, 0, s], [0, 1, 0], [-s, 0, c]])


def rotz_np(t):
    c, s = np.cos(t), np.sin(t)
    return np.array([[c, -s, 0], [s, c, 0], [0, 0, 1]])


def pose_from_oxts_packet_np(raw_data, scale):
    """(R [3,3], t [3]) of one OXTS packet (pose_utils.py:L46-81): Mercator projection of (lat, lon) scaled by `scale` = cos(lat of the drive's
    first frame), altitude as z, R = Rz(yaw) Ry(pitch) Rx(roll)."""
    if len(raw_data) != len(OXTS_FIELDS):
        raise ValueError(f"an OXTS packet has {len(OXTS_FIELDS)} fields, got {len(raw_data)}")
    lat, lon, alt, roll, pitch, yaw = (raw_data[i] for i in range(6))
    tx = scale * lon * np.pi * EARTH_RADIUS_M / 180.0
    ty = scale * EARTH_RADIUS_M * np.log(np.tan((90.0 + lat) * np.pi / 360.0))
    t = np.array([tx, ty, alt])
    R = rotz_np(yaw).dot(roty_np(pitch).dot(rotx_np(roll)))
    return R, t


def T_from_R_t_np(R, t):
    return np.vstack((np.hstack([np.reshape(R, (3, 3)), np.reshape(t, (3, 1))]), [0, 0, 0, 1]))


def invert_pose_np(T):
    """Inverse of a [4,4] rigid transform: [R^T | -R^T t]."""
    out = np.eye(4, dtype=T.dtype)
    out[:3, :3] = T[:3, :3].T
    out[:3, 3] = -(out[:3, :3] @ T[:3, 3])
    return out


def euler2mat(angle):
    """[B,3] Euler angles -> [B,3,3] rotation (pose_utils.py:L98-127): the rotation block of pose_vec2mat with a zero translation."""
    vec = torch.cat([torch.zeros_like(angle), angle], 1)
    return pose_vec2mat(vec)[:, :3, :3]


def invert_pose(T):
    """Inverse of [B,4,4] rigid transforms (pose_utils.py:L140-145)."""
    Rt = T[:, :3, :3].transpose(-2, -1)
    top = torch.cat([Rt, -(Rt @ T[:, :3, 3:])], 2)
    return torch.cat([top, T.new_tensor([0.0, 0.0, 0.0, 1.0]).expand(len(T), 1, 4)], 1)
