"""Parameter-holding building blocks whose forward runs on libsde_hip.so (NHWC activations).

Parameter names/layouts equal torch's (``weight`` OIHW fp32, ``bias``, BatchNorm ``running_mean`` ...), so the reference's
checkpoints load with ``load_state_dict`` unchanged.
"""
import math

import torch
import torch.nn as nn

from ..hip import nn as HN


class HipConv2d(nn.Module):
    """nn.Conv2d stand-in (weights only); the arithmetic is sde_conv_fwd / sde_conv_wgrad (include/sde_hip.h)."""

    def __init__(self, in_channels, out_channels, kernel_size, stride=1, padding=0, bias=True, reflect=False):
        super().__init__()
        self.in_channels, self.out_channels, self.kernel_size = int(in_channels), int(out_channels), int(kernel_size)
        self.stride, self.padding, self.reflect = int(stride), int(padding), bool(reflect)
        self.weight = nn.Parameter(torch.empty(self.out_channels, self.in_channels, self.kernel_size, self.kernel_size))
        self.bias = nn.Parameter(torch.empty(self.out_channels)) if bias else None
        self._packed = None        # (forward operand, dgrad operand) maintained by hip.nn.WeightPacker, else packed per call
        self._pack_shapes = None
        self.reset_parameters()

    def reset_parameters(self):   # torch.nn.Conv2d default initialisation
        nn.init.kaiming_uniform_(self.weight, a=math.sqrt(5))
        if self.bias is not None:
            bound = 1.0 / math.sqrt(self.in_channels * self.kernel_size ** 2)
            nn.init.uniform_(self.bias, -bound, bound)

    def forward(self, x, skip=None, upsample=False, act=HN.ACT_NONE, bn_stats=False, n_out=1):
        return HN.conv2d(x, self.weight, self.bias, self.stride, self.padding, self.reflect, act, skip, upsample, bn_stats, owner=self, n_out=n_out)

    def extra_repr(self):
        return f"{self.in_channels}, {self.out_channels}, k={self.kernel_size}, s={self.stride}, p={self.padding}, reflect={self.reflect}"


class HipBatchNorm2d(nn.Module):
    """nn.BatchNorm2d stand-in: batch statistics come from the producing convolution's epilogue (stats slab)."""

    def __init__(self, num_features, eps=1e-5, momentum=0.1):
        super().__init__()
        self.num_features, self.eps, self.momentum = int(num_features), eps, momentum
        self.weight = nn.Parameter(torch.ones(num_features))
        self.bias = nn.Parameter(torch.zeros(num_features))
        self.register_buffer("running_mean", torch.zeros(num_features))
        self.register_buffer("running_var", torch.ones(num_features))
        self.register_buffer("num_batches_tracked", torch.tensor(0, dtype=torch.long))
        self._pending_batches = 0     # counted on the host, folded into the buffer when the state dict is read (no per-layer kernel)

    def flush_counters(self):
        if self._pending_batches:
            self.num_batches_tracked += self._pending_batches
            self._pending_batches = 0

    def _save_to_state_dict(self, destination, prefix, keep_vars):
        self.flush_counters()
        super()._save_to_state_dict(destination, prefix, keep_vars)

    def _load_from_state_dict(self, *args, **kwargs):
        self._pending_batches = 0                  # the loaded num_batches_tracked is the whole count
        super()._load_from_state_dict(*args, **kwargs)

    def forward(self, y, stats, residual=None, relu=True, n_out=1):
        if self.training:
            self._pending_batches += 1
        return HN.batch_norm_act(y, stats, self.weight, self.bias, self.running_mean, self.running_var, residual, relu, self.momentum, self.eps,
                                 self.training, n_out)


class HipGroupNorm(nn.Module):
    """nn.GroupNorm stand-in fused with the following activation: ReLU (PoseNet.py:L13-20), "elu" (layers01.py:L33-40) or none."""

    def __init__(self, num_groups, num_channels, eps=1e-5):
        super().__init__()
        self.num_groups, self.num_channels, self.eps = num_groups, num_channels, eps
        self.weight = nn.Parameter(torch.ones(num_channels))
        self.bias = nn.Parameter(torch.zeros(num_channels))

    def forward(self, x, relu=True, residual=None):
        return HN.group_norm_relu(x, self.weight, self.bias, self.num_groups, self.eps, relu, residual)


def conv_bn(conv, bn, x, residual=None, relu=True, n_out=1):
    """conv -> training-mode BatchNorm [-> + residual] [-> ReLU]; in eval mode BN uses its running statistics.
    n_out > 1: that many aliases of the result, one per consumer (see hip.nn._BatchNormAct)."""
    if bn.training:
        y, stats = conv(x, bn_stats=True)
    else:
        y, stats = conv(x), None
    return bn(y, stats, residual, relu, n_out)
