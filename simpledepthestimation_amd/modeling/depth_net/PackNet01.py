"""PackNet01 on the HIP kernels.  Contract: detectron2/modeling/depth_net/PackNet01.py:L18-209 (its layers: layers/layers01.py).

``cfg.MODEL.DEPTH_NET.VERSION`` selects "1A" (skips concatenated) or "1B" (skips added, half-width upper decoder).  Module names -- and so
the state-dict keys -- are the reference's (``pre_calc``, ``conv1..5``, ``pack1..5``, ``unpack5..1``, ``iconv5..1``, ``dispN_layer``);
they are created from the width tables below instead of being spelled out one by one.  Batch contract: reads ``depth_net_input`` (or the
already laid-out ``depth_net_input_nhwc``), writes ``depth_pred`` = four [B,1,h,w] fp32 metric depth maps, finest first, obtained as
``disp_to_depth(sigmoid(.)/0.5)`` exactly like L199; ``UPSAMPLE_DEPTH`` brings all of them to the input resolution (L203-204).
"""
import torch
import torch.nn.functional as F
from torch import nn

from ...hip import nn as HN
from ...layers import layers01 as L01
from ...layers.depth_decoder import disp_to_depth
from .build import DEPTH_NET_REGISTRY
from .DepthResNet import compute_dtype

WIDTH = {0: 64, 1: 64, 2: 64, 3: 128, 4: 256, 5: 512}        # level 0 = pre_calc output, levels 1..5 = encoder stages
RES_BLOCKS = {2: 2, 3: 2, 4: 3, 5: 3}                        # residual blocks of conv2..conv5
PACK_KERNEL = {1: 5, 2: 3, 3: 3, 4: 3, 5: 3}
DISP_LEVELS = (1, 2, 3, 4)                                   # levels that emit an inverse-depth map
MIN_DEPTH = 0.1


def _decoder_widths(version):
    """Per level: (unpack output width, iconv input width).  Levels 1..3 also take the up-sampled disparity of the level below them."""
    table = {}
    for lvl in range(1, 6):
        w, skip_w = WIDTH[lvl], WIDTH[lvl - 1]
        extra = 1 if lvl <= 3 else 0
        if version == "A":
            table[lvl] = (w, w + skip_w + extra)
        else:
            half = w // 2 if lvl >= 3 else w
            table[lvl] = (half, half + extra)
    return table


@DEPTH_NET_REGISTRY.register()
class PackNet01(nn.Module):
    def __init__(self, cfg, **kwargs):
        nn.Module.__init__(self)
        self.version = cfg.MODEL.DEPTH_NET.VERSION[1:]
        if self.version not in ("A", "B"):
            raise ValueError("Unknown MonoDepth2 version {}".format(self.version))
        # registration order = the reference's (L60-100), so that parameter indices of an optimizer checkpoint line up
        self.pre_calc = L01.Conv2D(3, WIDTH[0], 5, 1)
        for lvl, k in PACK_KERNEL.items():
            setattr(self, f"pack{lvl}", L01.PackLayerConv3d(WIDTH[lvl], k))
        self.conv1 = L01.Conv2D(WIDTH[0], WIDTH[1], 7, 1)
        for lvl, blocks in RES_BLOCKS.items():
            setattr(self, f"conv{lvl}", L01.ResidualBlock(WIDTH[lvl - 1], WIDTH[lvl], blocks, 1, dropout=0.0))
        dec = _decoder_widths(self.version)
        for lvl in range(5, 0, -1):                                # unpack input: iconv of the level below (pack5 at the bottom)
            setattr(self, f"unpack{lvl}", L01.UnpackLayerConv3d(WIDTH[min(lvl + 1, 5)], dec[lvl][0], 3))
        for lvl in range(5, 0, -1):
            setattr(self, f"iconv{lvl}", L01.Conv2D(dec[lvl][1], WIDTH[lvl], 3, 1))
        for lvl in reversed(DISP_LEVELS):
            setattr(self, f"disp{lvl}_layer", L01.InvDepth(WIDTH[lvl], out_channels=1))
        self.max_depth = cfg.MODEL.MAX_DEPTH
        self.upsample_depth = cfg.MODEL.DEPTH_NET.UPSAMPLE_DEPTH
        self.dtype = compute_dtype(cfg)
        # no _grad_cut attribute: HipTrainer then all-reduces after the full backward (no two-phase overlap for this net)

    def scale_inv_depth(self, disp):
        return disp_to_depth(disp, min_depth=MIN_DEPTH, max_depth=self.max_depth)

    def _join(self, unpacked, skip, disp_below=None):
        """The input of iconv_l (PackNet01.py:L150-199): cat([unpacked, skip(, nearest_x2(inverse depth of the level below))]) in version A,
        [unpacked + skip(, ...)] in version B -- one kernel (sde_concat_fwd), zero-filled to the 16-byte group."""
        return HN.concat(unpacked, skip, disp_below, add=self.version != "A")

    def forward(self, batch):
        flip = bool(batch.get("flip", False))
        stem = batch.get("depth_net_input_nhwc")
        if stem is None:
            stem = HN.prep_input(batch["depth_net_input"], None, None, self.dtype, flip)       # flip folded into the layout change
        stem = self.pre_calc(stem)
        # encoder: packed[l] is the output of level l after its space-to-depth packing; packed[0] is the full-resolution stem
        packed = {0: stem}
        for lvl in range(1, 6):
            feat = getattr(self, f"conv{lvl}")(packed[lvl - 1])
            packed[lvl] = getattr(self, f"pack{lvl}")(feat)
        # decoder, coarse to fine; the inverse depth of level l+1 is an extra input channel of levels 3..1
        feat, disp, dmap = packed[5], {}, {}
        for lvl in range(5, 0, -1):
            joined = self._join(getattr(self, f"unpack{lvl}")(feat), packed[lvl - 1], disp.get(lvl + 1))
            feat = getattr(self, f"iconv{lvl}")(joined)
            if lvl in DISP_LEVELS:
                # (inverse depth for the level above, metric depth = scale_inv_depth(.)[1] flipped back): one kernel
                disp[lvl], dmap[lvl] = getattr(self, f"disp{lvl}_layer")(feat, MIN_DEPTH, self.max_depth, flip)
        depth = [dmap[lvl] for lvl in DISP_LEVELS]
        if self.upsample_depth:
            depth = [F.interpolate(d.contiguous(), size=tuple(stem.shape[1:3]), mode="nearest") for d in depth]
        batch["depth_pred"] = depth
        return batch
