"""Pose-net registry (reference: detectron2/modeling/pose_net/build.py:L5-31)."""
import torch.nn as nn

from ...utils.registry import Registry

POSE_NET_REGISTRY = Registry("POSE_NET")


def build_pose_net(cfg, input_shape=None):
    pose_net = POSE_NET_REGISTRY.get(cfg.MODEL.POSE_NET.NAME)(cfg)
    assert isinstance(pose_net, nn.Module)
    return pose_net
