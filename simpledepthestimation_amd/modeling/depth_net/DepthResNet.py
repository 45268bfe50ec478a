"""DepthResNet = ResnetEncoder + DepthDecoder + disp_to_depth (reference: detectron2/modeling/depth_net/DepthResNet.py:L15-70)."""
import torch
import torch.nn as nn
import torch.nn.functional as F

from ...hip import nn as HN
from ...hip import photometric as HP
from ...layers.depth_decoder import DepthDecoder
from ...layers.resnet_encoder import ResnetEncoder
from .build import DEPTH_NET_REGISTRY

_DTYPES = {"fp32": torch.float32, "float32": torch.float32, "bf16": torch.bfloat16, "bfloat16": torch.bfloat16,
           "fp16": torch.float16, "float16": torch.float16}      # fp16: with SOLVER.AMP (dynamic loss scaling), BASELINE.json configs[4]


def compute_dtype(cfg):
    name = str(cfg.MODEL.get("COMPUTE_DTYPE", "fp32")) if hasattr(cfg.MODEL, "get") else str(getattr(cfg.MODEL, "COMPUTE_DTYPE", "fp32"))
    if name not in _DTYPES:
        raise ValueError(f"MODEL.COMPUTE_DTYPE must be one of {sorted(_DTYPES)}, got {name}")
    return _DTYPES[name]


@DEPTH_NET_REGISTRY.register()
class DepthResNet(nn.Module):
    def __init__(self, cfg, **kwargs):
        super().__init__()
        version = cfg.MODEL.DEPTH_NET.ENCODER_NAME
        assert version is not None, "DispResNet needs a version"
        num_layers = int(version[:2])
        pretrained = version[2:] == "pt"
        assert num_layers in [18, 34, 50], "ResNet version {} not available".format(num_layers)
        self.encoder = ResnetEncoder(num_layers=num_layers, pretrained=pretrained)
        self.decoder = DepthDecoder(num_ch_enc=self.encoder.num_ch_enc)
        self.min_depth, self.max_depth = 0.1, float(cfg.MODEL.MAX_DEPTH)
        self.upsample_depth = cfg.MODEL.DEPTH_NET.UPSAMPLE_DEPTH
        self.dtype = compute_dtype(cfg)
        # optional callable installed by engine.trainer.HipTrainer(overlap=True): splits the autograd graph at the encoder features
        # so that the backward of decoder/layer4/layer3 (most parameters) finishes first and its all-reduce overlaps the rest
        self._grad_cut = None

    def forward(self, batch):
        """Consumes batch['depth_net_input'] ([B,3,H,W] normalised, as in the reference) or the fused NHWC form the meta-archs of
        this package provide under batch['depth_net_input_nhwc']; adds depth_pred (4 x [B,1,h,w] fp32) and res2/3/4."""
        flip = bool(batch.get("flip", False))
        x = batch.get("depth_net_input_nhwc")
        if x is None:
            x = HN.prep_input(batch["depth_net_input"], None, None, self.dtype, flip)   # flip folded into the layout change
        cut = self._grad_cut if (self._grad_cut is not None and torch.is_grad_enabled()) else None
        feats = self.encoder(x, cut=cut)
        logits = self.decoder(feats)
        disps = [HN.depth_head(logits[("disp_logit", i)], self.min_depth, self.max_depth, flip) for i in range(4)]
        if self.upsample_depth:
            # DepthResNet.py:L62-63: resize_img(d, input size, mode='nearest') -- a 1-channel nearest up-sampling whose backward is the sum over
            # each replicated block: torch's own device op (F.interpolate), exactly what the reference calls
            size = tuple(x.shape[1:3])
            disps = [F.interpolate(d, size=size, mode="nearest") for d in disps]
        batch.update({"res2": disps[3], "res3": disps[2], "res4": disps[1], "depth_pred": disps})
        return batch
