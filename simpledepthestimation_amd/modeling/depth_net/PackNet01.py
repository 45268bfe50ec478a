"""PackNet01 (reference: detectron2/modeling/depth_net/PackNet01.py:L18-209; layers in layers/layers01.py) on the HIP kernels.

Same constructor contract (``cfg.MODEL.DEPTH_NET.VERSION`` = "1A" concatenation / "1B" addition), same module tree and state-dict
keys, same batch contract: consumes ``depth_net_input`` (or the fused NHWC form), adds ``depth_pred`` = 4 x [B,1,h,w] fp32 metric depth,
``disp_to_depth`` applied on top of sigmoid/0.5 as the reference does (PackNet01.py:L199).
"""
from functools import partial

import torch
import torch.nn as nn
import torch.nn.functional as F

from ...hip import nn as HN
from ...layers.depth_decoder import disp_to_depth
from ...layers.layers01 import Conv2D, InvDepth, PackLayerConv3d, ResidualBlock, UnpackLayerConv3d
from .build import DEPTH_NET_REGISTRY
from .DepthResNet import compute_dtype


def _cat(parts, vec):
    """Channel concatenation of NHWC tensors, zero-padded to the next multiple of the 16-byte group."""
    c = sum(p.shape[-1] for p in parts)
    pad = (-c) % vec
    if pad:
        B, H, W, _ = parts[0].shape
        parts = list(parts) + [torch.zeros(B, H, W, pad, device=parts[0].device, dtype=parts[0].dtype)]
    return torch.cat(parts, -1)


def _up2(disp):
    """nn.Upsample(scale_factor=2, mode='nearest') of a [B,H,W] map -> [B,2H,2W]."""
    return disp.repeat_interleave(2, 1).repeat_interleave(2, 2)


@DEPTH_NET_REGISTRY.register()
class PackNet01(nn.Module):
    def __init__(self, cfg, **kwargs):
        super().__init__()
        self.version = cfg.MODEL.DEPTH_NET.VERSION[1:]
        in_channels, out_channels = 3, 1
        ni, no = 64, out_channels
        n1, n2, n3, n4, n5 = 64, 64, 128, 256, 512
        num_blocks = [2, 2, 3, 3]
        pack_kernel = [5, 3, 3, 3, 3]
        unpack_kernel = [3, 3, 3, 3, 3]
        iconv_kernel = [3, 3, 3, 3, 3]
        self.pre_calc = Conv2D(in_channels, ni, 5, 1)
        if self.version == "A":        # channel concatenation
            n1o, n1i = n1, n1 + ni + no
            n2o, n2i = n2, n2 + n1 + no
            n3o, n3i = n3, n3 + n2 + no
            n4o, n4i = n4, n4 + n3
            n5o, n5i = n5, n5 + n4
        elif self.version == "B":      # channel addition
            n1o, n1i = n1, n1 + no
            n2o, n2i = n2, n2 + no
            n3o, n3i = n3 // 2, n3 // 2 + no
            n4o, n4i = n4 // 2, n4 // 2
            n5o, n5i = n5 // 2, n5 // 2
        else:
            raise ValueError("Unknown MonoDepth2 version {}".format(self.version))
        # encoder
        self.pack1 = PackLayerConv3d(n1, pack_kernel[0])
        self.pack2 = PackLayerConv3d(n2, pack_kernel[1])
        self.pack3 = PackLayerConv3d(n3, pack_kernel[2])
        self.pack4 = PackLayerConv3d(n4, pack_kernel[3])
        self.pack5 = PackLayerConv3d(n5, pack_kernel[4])
        self.conv1 = Conv2D(ni, n1, 7, 1)
        self.conv2 = ResidualBlock(n1, n2, num_blocks[0], 1, dropout=0.0)
        self.conv3 = ResidualBlock(n2, n3, num_blocks[1], 1, dropout=0.0)
        self.conv4 = ResidualBlock(n3, n4, num_blocks[2], 1, dropout=0.0)
        self.conv5 = ResidualBlock(n4, n5, num_blocks[3], 1, dropout=0.0)
        # decoder
        self.unpack5 = UnpackLayerConv3d(n5, n5o, unpack_kernel[0])
        self.unpack4 = UnpackLayerConv3d(n5, n4o, unpack_kernel[1])
        self.unpack3 = UnpackLayerConv3d(n4, n3o, unpack_kernel[2])
        self.unpack2 = UnpackLayerConv3d(n3, n2o, unpack_kernel[3])
        self.unpack1 = UnpackLayerConv3d(n2, n1o, unpack_kernel[4])
        self.iconv5 = Conv2D(n5i, n5, iconv_kernel[0], 1)
        self.iconv4 = Conv2D(n4i, n4, iconv_kernel[1], 1)
        self.iconv3 = Conv2D(n3i, n3, iconv_kernel[2], 1)
        self.iconv2 = Conv2D(n2i, n2, iconv_kernel[3], 1)
        self.iconv1 = Conv2D(n1i, n1, iconv_kernel[4], 1)
        # depth layers (the reference's parameter-free unpack_disp* upsamplers are _up2 here)
        self.disp4_layer = InvDepth(n4, out_channels=out_channels)
        self.disp3_layer = InvDepth(n3, out_channels=out_channels)
        self.disp2_layer = InvDepth(n2, out_channels=out_channels)
        self.disp1_layer = InvDepth(n1, out_channels=out_channels)
        self.scale_inv_depth = partial(disp_to_depth, min_depth=0.1, max_depth=cfg.MODEL.MAX_DEPTH)
        self.upsample_depth = cfg.MODEL.DEPTH_NET.UPSAMPLE_DEPTH
        self.dtype = compute_dtype(cfg)
        # no _grad_cut attribute: HipTrainer then all-reduces after the full backward (no two-phase overlap for this net)

    def _join(self, unpack, skip, udisp=None):
        vec = 8 if self.dtype == torch.bfloat16 else 4
        parts = [unpack, skip] if self.version == "A" else [unpack + skip]
        if udisp is not None:
            parts.append(udisp.unsqueeze(-1).to(self.dtype))
        return parts[0] if len(parts) == 1 else _cat(parts, vec)

    def forward(self, batch):
        flip = bool(batch.get("flip", False))
        x = batch.get("depth_net_input_nhwc")
        if x is None:
            x = HN.prep_input(batch["depth_net_input"], None, None, self.dtype, flip)       # flip folded into the layout change
        x = self.pre_calc(x)
        # encoder
        x1 = self.conv1(x)
        x1p = self.pack1(x1)
        x2 = self.conv2(x1p)
        x2p = self.pack2(x2)
        x3 = self.conv3(x2p)
        x3p = self.pack3(x3)
        x4 = self.conv4(x3p)
        x4p = self.pack4(x4)
        x5 = self.conv5(x4p)
        x5p = self.pack5(x5)
        skip1, skip2, skip3, skip4, skip5 = x, x1p, x2p, x3p, x4p
        # decoder
        iconv5 = self.iconv5(self._join(self.unpack5(x5p), skip5))
        iconv4 = self.iconv4(self._join(self.unpack4(iconv5), skip4))
        disp4 = self.disp4_layer(iconv4)
        iconv3 = self.iconv3(self._join(self.unpack3(iconv4), skip3, _up2(disp4)))
        disp3 = self.disp3_layer(iconv3)
        iconv2 = self.iconv2(self._join(self.unpack2(iconv3), skip2, _up2(disp3)))
        disp2 = self.disp2_layer(iconv2)
        iconv1 = self.iconv1(self._join(self.unpack1(iconv2), skip1, _up2(disp2)))
        disp1 = self.disp1_layer(iconv1)
        disps = [self.scale_inv_depth(d.unsqueeze(1))[1] for d in (disp1, disp2, disp3, disp4)]
        if flip:
            disps = [torch.flip(d, [3]) for d in disps]
        if self.upsample_depth:       # PackNet01.py:L203-204
            disps = [F.interpolate(d.contiguous(), size=tuple(x.shape[1:3]), mode="nearest") for d in disps]
        batch["depth_pred"] = disps
        return batch
