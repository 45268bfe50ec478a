"""MonoDepth2-style depth decoder on the HIP convolution engine (contract: detectron2/layers/depth_decoder.py:L9-110).

State-dict layout is the reference's: one ``decoder`` ModuleList holding, in this order, the two ELU convolutions of every level from the
coarsest (4) to the finest (0), then the four disparity heads; ELU convolutions keep their parameters at ``decoder.K.conv.conv.*`` and
heads at ``decoder.K.conv.*``.  What the reference does as separate modules -- reflection padding, the nearest x2 up-sampling, the skip
concatenation, the ELU -- is folded into the convolution's loader / epilogue here, so a level is two kernel launches.
"""
import torch
from torch import nn

from ..hip import nn as HN
from .hip_modules import HipConv2d

DEC_WIDTH = (16, 32, 64, 128, 256)        # decoder width per level, finest first


def disp_to_depth(disp, min_depth, max_depth):
    """Sigmoid disparity -> (scaled disparity, depth) between the two depth bounds (depth_decoder.py:L9-18).  Torch tensors; the training
    path uses the fused ``sde_depth_head_*`` kernels instead."""
    lo, hi = 1.0 / max_depth, 1.0 / min_depth
    scaled = lo + (hi - lo) * disp
    return scaled, 1.0 / scaled


class Conv3x3(nn.Module):
    """3x3 convolution over a reflection- (or zero-) padded input (depth_decoder.py:L36-53)."""

    def __init__(self, in_channels, out_channels, use_refl=True):
        nn.Module.__init__(self)
        self.conv = HipConv2d(int(in_channels), int(out_channels), 3, 1, 1, bias=True, reflect=use_refl)

    def forward(self, x, skip=None, upsample=False, act=HN.ACT_NONE, n_out=1):
        return self.conv(x, skip=skip, upsample=upsample, act=act, n_out=n_out)


class ConvBlock(nn.Module):
    """Conv3x3 followed by ELU (depth_decoder.py:L21-33); the activation runs in the convolution's epilogue."""

    def __init__(self, in_channels, out_channels):
        nn.Module.__init__(self)
        self.conv = Conv3x3(in_channels, out_channels)

    def forward(self, x, skip=None, upsample=False, n_out=1):
        return self.conv(x, skip=skip, upsample=upsample, act=HN.ACT_ELU, n_out=n_out)


class DepthDecoder(nn.Module):
    def __init__(self, num_ch_enc, scales=range(4), num_output_channels=1, use_skips=True):
        nn.Module.__init__(self)
        self.num_ch_enc = num_ch_enc
        self.scales = scales
        self.num_output_channels = num_output_channels
        self.use_skips = use_skips
        top = len(DEC_WIDTH) - 1
        blocks, self._slot = [], {}
        below = int(num_ch_enc[-1])
        for lvl in range(top, -1, -1):
            width = DEC_WIDTH[lvl]
            skip_w = int(num_ch_enc[lvl - 1]) if (use_skips and lvl > 0) else 0
            self._slot["reduce", lvl] = len(blocks)
            blocks.append(ConvBlock(below, width))
            self._slot["merge", lvl] = len(blocks)
            blocks.append(ConvBlock(width + skip_w, width))
            below = width
        for s in scales:
            self._slot["head", s] = len(blocks)
            blocks.append(Conv3x3(DEC_WIDTH[s], num_output_channels))
        self.decoder = nn.ModuleList(blocks)

    def _block(self, kind, lvl):
        return self.decoder[self._slot[kind, lvl]]

    def forward(self, input_features):
        """Returns {("disp_logit", i): NHWC tensor whose channel 0 is the pre-activation disparity} for every scale."""
        outputs = {}
        feat = input_features[-1]
        for lvl in range(len(DEC_WIDTH) - 1, -1, -1):
            feat = self._block("reduce", lvl)(feat)
            skip = input_features[lvl - 1] if (self.use_skips and lvl > 0) else None
            # a level with a disparity head that is not the last one has two consumers: two aliases, so that backward sums the two gradients
            # inside the ELU-backward kernel instead of through an autograd add
            two = lvl in self.scales and lvl > 0 and torch.is_grad_enabled()
            feat = self._block("merge", lvl)(feat, skip=skip, upsample=True, n_out=2 if two else 1)    # up-sampling + concatenation happen in the loader
            head_in = feat
            if two:
                feat, head_in = feat
            if lvl in self.scales:
                outputs["disp_logit", lvl] = self._block("head", lvl)(head_in)
        self.outputs = outputs
        return outputs
