"""MonoDepth2-style depth decoder on the HIP convolution engine.

Reference: detectron2/layers/depth_decoder.py:L9-110.  ModuleList order (= state-dict indices) is the reference's:
upconv(4,0),(4,1),(3,0),...,(0,1), dispconv 0..3; ConvBlock parameters live at ``decoder.K.conv.conv.{weight,bias}``,
dispconv at ``decoder.K.conv.{weight,bias}``.  Reflection padding, the nearest x2 upsample and the skip concat are folded into
the convolution's loader; ELU into its epilogue.
"""
from collections import OrderedDict

import numpy as np
import torch.nn as nn

from ..hip import nn as HN
from .hip_modules import HipConv2d


def disp_to_depth(disp, min_depth, max_depth):
    """depth_decoder.py:L9-18 (torch tensors; the training path uses the fused sde_depth_head_* kernels instead)."""
    min_disp = 1 / max_depth
    max_disp = 1 / min_depth
    scaled_disp = min_disp + (max_disp - min_disp) * disp
    depth = 1 / scaled_disp
    return scaled_disp, depth


class Conv3x3(nn.Module):
    """ReflectionPad2d(1) + Conv2d(3x3) (depth_decoder.py:L36-53)."""

    def __init__(self, in_channels, out_channels, use_refl=True):
        super().__init__()
        self.conv = HipConv2d(int(in_channels), int(out_channels), 3, 1, 1, bias=True, reflect=use_refl)

    def forward(self, x, skip=None, upsample=False, act=HN.ACT_NONE):
        return self.conv(x, skip=skip, upsample=upsample, act=act)


class ConvBlock(nn.Module):
    """Conv3x3 + ELU (depth_decoder.py:L21-33)."""

    def __init__(self, in_channels, out_channels):
        super().__init__()
        self.conv = Conv3x3(in_channels, out_channels)

    def forward(self, x, skip=None, upsample=False):
        return self.conv(x, skip=skip, upsample=upsample, act=HN.ACT_ELU)


class DepthDecoder(nn.Module):
    def __init__(self, num_ch_enc, scales=range(4), num_output_channels=1, use_skips=True):
        super().__init__()
        self.num_output_channels, self.use_skips, self.scales = num_output_channels, use_skips, scales
        self.num_ch_enc = num_ch_enc
        self.num_ch_dec = np.array([16, 32, 64, 128, 256])
        self.convs = OrderedDict()
        for i in range(4, -1, -1):
            num_ch_in = self.num_ch_enc[-1] if i == 4 else self.num_ch_dec[i + 1]
            self.convs[("upconv", i, 0)] = ConvBlock(num_ch_in, self.num_ch_dec[i])
            num_ch_in = self.num_ch_dec[i]
            if self.use_skips and i > 0:
                num_ch_in += self.num_ch_enc[i - 1]
            self.convs[("upconv", i, 1)] = ConvBlock(num_ch_in, self.num_ch_dec[i])
        for s in self.scales:
            self.convs[("dispconv", s)] = Conv3x3(self.num_ch_dec[s], self.num_output_channels)
        self.decoder = nn.ModuleList(list(self.convs.values()))

    def forward(self, input_features):
        """Returns {("disp_logit", i): NHWC tensor whose channel 0 is the pre-softplus disparity} for the 4 scales."""
        self.outputs = {}
        x = input_features[-1]
        for i in range(4, -1, -1):
            x = self.convs[("upconv", i, 0)](x)
            skip = input_features[i - 1] if (self.use_skips and i > 0) else None
            x = self.convs[("upconv", i, 1)](x, skip=skip, upsample=True)     # upsample + cat happen in the loader
            if i in self.scales:
                self.outputs[("disp_logit", i)] = self.convs[("dispconv", i)](x)
        return self.outputs
