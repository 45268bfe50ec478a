"""Depth-net registry (reference: detectron2/modeling/depth_net/build.py:L5-31)."""
import torch.nn as nn

from ...utils.registry import Registry

DEPTH_NET_REGISTRY = Registry("DEPTH_NET")


def build_depth_net(cfg, input_shape=None):
    depth_net = DEPTH_NET_REGISTRY.get(cfg.MODEL.DEPTH_NET.NAME)(cfg)
    assert isinstance(depth_net, nn.Module)
    return depth_net
