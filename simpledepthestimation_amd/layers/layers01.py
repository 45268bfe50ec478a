"""PackNet building blocks on the HIP kernels (reference: detectron2/layers/layers01.py:L11-298), NHWC activations.

Module / parameter names equal the reference's (``conv_base``, ``normalize``, ``conv3d`` ...), so its checkpoints load unchanged.
Everything between the convolutions runs in libsde_hip.so as well (csrc/packnet.hip): space-to-depth ``packing`` and ``nn.PixelShuffle`` are one
16-byte-per-lane permutation kernel each (each other's backward), the residual sum lives inside GroupNorm, the inverse-depth head is one kernel.
"""
import torch
import torch.nn as nn

from ..hip import nn as HN
from .hip_modules import HipConv2d, HipGroupNorm


def _xavier_(conv):
    nn.init.xavier_uniform_(conv.weight)         # PackNet01.init_weights (PackNet01.py:L110-116)
    if conv.bias is not None:
        nn.init.zeros_(conv.bias)


class Conv2D(nn.Module):
    """ConstantPad2d(k//2) + nn.Conv2d(k, stride) + GroupNorm(16) + ELU (layers01.py:L11-41)."""

    def __init__(self, in_channels, out_channels, kernel_size, stride):
        super().__init__()
        self.kernel_size = kernel_size
        self.conv_base = HipConv2d(in_channels, out_channels, kernel_size, stride=stride, padding=kernel_size // 2, bias=True)
        self.normalize = HipGroupNorm(16, out_channels)
        _xavier_(self.conv_base)

    def forward(self, x):
        return self.normalize(self.conv_base(x), relu="elu")


class ResidualConv(nn.Module):
    """layers01.py:L44-77 (dropout is 0.0 in PackNet01)."""

    def __init__(self, in_channels, out_channels, stride, dropout=None):
        super().__init__()
        if dropout:
            raise NotImplementedError("ResidualConv dropout > 0 (PackNet01 passes 0.0)")
        self.conv1 = Conv2D(in_channels, out_channels, 3, stride)
        self.conv2 = Conv2D(out_channels, out_channels, 3, 1)
        self.conv3 = HipConv2d(in_channels, out_channels, 1, stride=stride, padding=0, bias=True)
        self.normalize = HipGroupNorm(16, out_channels)
        _xavier_(self.conv3)

    def forward(self, x):
        x_out = self.conv2(self.conv1(x))
        shortcut = self.conv3(x)
        return self.normalize(x_out, relu="elu", residual=shortcut)      # normalize(x_out + shortcut): the sum is formed inside the GroupNorm kernels


def ResidualBlock(in_channels, out_channels, num_blocks, stride, dropout=None):
    """layers01.py:L80-102."""
    layers = [ResidualConv(in_channels, out_channels, stride, dropout=dropout)]
    for _ in range(1, num_blocks):
        layers.append(ResidualConv(out_channels, out_channels, 1, dropout=dropout))
    return nn.Sequential(*layers)


class InvDepth(nn.Module):
    """ConstantPad2d(1) + Conv2d(3x3 -> 1) + Sigmoid, divided by min_depth (layers01.py:L105-133).  Returns [B,H,W] fp32."""

    def __init__(self, in_channels, out_channels=1, min_depth=0.5):
        super().__init__()
        if out_channels != 1:
            raise NotImplementedError("InvDepth with out_channels != 1")
        self.min_depth = min_depth
        self.conv1 = HipConv2d(in_channels, out_channels, 3, stride=1, padding=1, bias=True)
        _xavier_(self.conv1)

    def forward(self, x, min_depth=0.1, max_depth=80.0, flip=False):
        """Returns (inverse depth [B,H,W] fp32 = sigmoid / min_depth of this head, metric depth [B,1,H,W] fp32 = PackNet01.scale_inv_depth(.)[1])."""
        return HN.inv_depth_head(self.conv1(x), self.min_depth, min_depth, max_depth, flip)


def packing(x, r=2):
    """Space-to-depth on NHWC: out[b,y,x, c*r*r + dy*r + dx] = in[b, y*r+dy, x*r+dx, c] (layers01.py:L138-160)."""
    if r != 2:
        raise NotImplementedError("packing with r != 2 (PackNet01 uses 2)")
    return HN.space_to_depth(x)


def pixel_shuffle(x, r=2):
    """nn.PixelShuffle(r) on NHWC: out[b, y*r+dy, x*r+dx, c] = in[b,y,x, c*r*r + dy*r + dx]."""
    if r != 2:
        raise NotImplementedError("pixel_shuffle with r != 2 (PackNet01 uses 2)")
    return HN.depth_to_space(x)


class _Conv3dParams(nn.Module):
    """Holds nn.Conv3d(1, d, 3, padding=1)'s parameters under the reference's names (``conv3d.weight`` [d,1,3,3,3], ``conv3d.bias``)."""

    def __init__(self, d):
        super().__init__()
        self.weight = nn.Parameter(torch.empty(d, 1, 3, 3, 3))
        self.bias = nn.Parameter(torch.zeros(d))
        nn.init.xavier_uniform_(self.weight)

    def forward(self, x):
        return HN.conv3d_pack(x, self.weight, self.bias)


class PackLayerConv3d(nn.Module):
    """packing -> Conv3d(1, d) over (channel, y, x) -> view(b, d*4C, h, w) -> Conv2D(4C*d -> C) (layers01.py:L223-259)."""

    def __init__(self, in_channels, kernel_size, r=2, d=8):
        super().__init__()
        if d != 8:
            raise NotImplementedError("PackLayerConv3d with d != 8")
        self.r = r
        self.conv = Conv2D(in_channels * (r ** 2) * d, in_channels, kernel_size, 1)
        self.conv3d = _Conv3dParams(d)

    def forward(self, x):
        return self.conv(self.conv3d(packing(x, self.r)))


class UnpackLayerConv3d(nn.Module):
    """Conv2D(Cin -> 4*Cout/d) -> Conv3d(1, d) -> view -> PixelShuffle(r) (layers01.py:L262-298)."""

    def __init__(self, in_channels, out_channels, kernel_size, r=2, d=8):
        super().__init__()
        if d != 8:
            raise NotImplementedError("UnpackLayerConv3d with d != 8")
        self.r = r
        self.conv = Conv2D(in_channels, out_channels * (r ** 2) // d, kernel_size, 1)
        self.conv3d = _Conv3dParams(d)

    def forward(self, x):
        return pixel_shuffle(self.conv3d(self.conv(x)), self.r)
