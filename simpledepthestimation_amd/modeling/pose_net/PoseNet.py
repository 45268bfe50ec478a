"""PoseNet on the HIP kernels (contract of detectron2/modeling/pose_net/PoseNet.py:L13-65): seven stride-2 convolutions, each followed by
GroupNorm(16) + ReLU, a 1x1 head with 6 outputs per context frame, the spatial mean of the head scaled by 0.01, and pose_vec2mat.

State-dict keys are the reference's: ``convN.0.{weight,bias}`` (the convolution) and ``convN.1.{weight,bias}`` (the GroupNorm) -- its ReLU at
index 2 has no parameters and is fused into the GroupNorm kernel here -- plus ``pose_pred.{weight,bias}``.  Initialisation: Xavier-uniform
convolution weights, zero biases (L42-48)."""
from torch import nn

from ...hip import nn as HN
from ...hip import photometric as HP
from ...layers.hip_modules import HipConv2d, HipGroupNorm
from ..depth_net.DepthResNet import compute_dtype
from .build import POSE_NET_REGISTRY

_STAGES = ((16, 7), (32, 5), (64, 3), (128, 3), (256, 3), (256, 3), (256, 3))        # (output channels, kernel size), all stride 2


@POSE_NET_REGISTRY.register()
class PoseNet(nn.Module):
    def __init__(self, cfg, **kwargs):
        nn.Module.__init__(self)
        self.nb_ref_imgs = int(cfg.MODEL.POSE_NET.NUM_CONTEXTS)
        width = 3 * (1 + self.nb_ref_imgs)
        for idx, (out_ch, k) in enumerate(_STAGES, start=1):
            conv = HipConv2d(width, out_ch, k, stride=2, padding=(k - 1) // 2, bias=True)
            setattr(self, f"conv{idx}", nn.Sequential(conv, HipGroupNorm(16, out_ch)))
            width = out_ch
        self.pose_pred = HipConv2d(width, 6 * self.nb_ref_imgs, 1, stride=1, padding=0, bias=True)
        self.dtype = compute_dtype(cfg)
        for mod in self.modules():
            if isinstance(mod, HipConv2d):
                nn.init.xavier_uniform_(mod.weight)
                nn.init.zeros_(mod.bias)

    def forward(self, batch):
        feat = batch.get("pose_net_input_nhwc")
        if feat is None:
            feat = HN.prep_input(batch["pose_net_input"], None, None, self.dtype)          # NCHW fp32 -> padded NHWC in the compute dtype
        for idx in range(1, len(_STAGES) + 1):
            conv, norm = getattr(self, f"conv{idx}")
            feat = norm(conv(feat))                                                      # GroupNorm + ReLU in one kernel
        head = self.pose_pred(feat)[..., : 6 * self.nb_ref_imgs]                         # [B, h, w, 6n] of the padded 16-byte groups
        vec = head.float().mean(dim=(1, 2)).mul(0.01).view(-1, self.nb_ref_imgs, 6)      # tiny reduction: torch glue
        batch["pose_vec"] = vec
        batch["pose_pred"] = [HP.pose_vec2mat(vec[:, j].contiguous()) for j in range(self.nb_ref_imgs)]
        return batch
