"""MonoDepth2Model (reference: detectron2/modeling/meta_arch/MonoDepth2.py:L21-151), intended semantics of SURVEY.md fact 4
(the translation is constant per sample).  The per-(scale, context) chain view_synthesis -> SSIM -> L1 mix -> min/auto-mask
is one fused kernel per scale; the image pyramid is built once per batch instead of once per (scale, context)."""
from collections import defaultdict

import torch
import torch.nn as nn

from ...hip import nn as HN
from ...hip import photometric as HP
from ...utils.memory import to_cuda
from ..depth_net import build_depth_net
from ..losses.losses import silog_loss, variance_loss
from ..losses.smoothness_loss import smoothness_loss
from ..losses.ssim_loss import SSIM
from ..pose_net import build_pose_net
from .build import META_ARCH_REGISTRY


@META_ARCH_REGISTRY.register()
class MonoDepth2Model(nn.Module):
    def __init__(self, cfg):
        super().__init__()
        self.depth_net = build_depth_net(cfg)
        self.pose_net = build_pose_net(cfg)
        self.ssim = SSIM(cfg.LOSS.C1, cfg.LOSS.C2)
        self.ssim_loss_weight = cfg.LOSS.SSIM_WEIGHT
        self.photometric_reduce = cfg.LOSS.PHOTOMETRIC_REDUCE
        self.use_automask = cfg.LOSS.AUTOMASK
        self.clip_loss = cfg.LOSS.CLIP
        self.var_loss_w = cfg.LOSS.VAR_LOSS_WEIGHT
        self.sup_loss_w = cfg.LOSS.SUPERVISED_WEIGHT
        self.smooth_loss_w = cfg.LOSS.SMOOTHNESS_WEIGHT
        self.supervise_loss = silog_loss(cfg.LOSS.VARIANCE_FOCUS)
        self.register_buffer("pixel_mean", torch.Tensor(cfg.MODEL.PIXEL_MEAN).view(1, -1, 1, 1))
        self.register_buffer("pixel_std", torch.Tensor(cfg.MODEL.PIXEL_STD).view(1, -1, 1, 1))

    @property
    def device(self):
        return self.pixel_mean.device

    def forward(self, batch):
        batch = to_cuda(batch, self.device)
        output = {}
        batch["depth_net_input_nhwc"] = HN.prep_input(batch["img"], self.pixel_mean, self.pixel_std, self.depth_net.dtype,
                                                      bool(batch.get("flip", False)))
        batch = self.depth_net(batch)
        if not self.training:
            output["depth_pred"] = batch["depth_pred"][0]
            return output
        batch["pose_net_input"] = torch.cat([batch["img"]] + batch["ctx_img"], 1)       # augmented frames, not normalised (L65)
        batch = self.pose_net(batch)
        image, contexts, intrinsics = batch["img_orig"], batch["ctx_img_orig"], batch["intrinsics"].float().contiguous()
        depth_pred, poses = batch["depth_pred"], batch["pose_pred"]
        num_scales = len(depth_pred)
        H, W = image.shape[-2:]
        losses = defaultdict(lambda: 0)
        photo_losses = []
        for i in range(num_scales):
            scale_w = 1.0 / 2 ** (num_scales - i - 1)
            h, w = depth_pred[i].shape[-2:]
            resized_image = HP.resize(image, (h, w))
            resized_targets = [HP.resize(c, (h, w)) for c in contexts]
            photo_losses.append(HP.photometric_scale_loss(depth_pred[i], intrinsics, resized_image, resized_targets, poses, w / W, h / H,
                                                          ssim_w=self.ssim_loss_weight, C1=self.ssim.C1, C2=self.ssim.C2,
                                                          automask=self.use_automask, reduce=self.photometric_reduce, clip=self.clip_loss))
            if self.smooth_loss_w > 0.0:
                losses["smooth_loss"] += smoothness_loss(depth_pred[i], resized_image) * (scale_w * self.smooth_loss_w / num_scales)
            if self.sup_loss_w > 0.0:
                # the reference weights this term with smooth_loss_w (MonoDepth2.py:L109, sic)
                losses["sup_loss"] += self.supervise_loss(depth_pred[i], batch["depth"]) * (scale_w * self.smooth_loss_w / num_scales)
            if self.var_loss_w > 0.0:
                losses["var_loss"] += variance_loss(depth_pred[i]) * (scale_w * self.var_loss_w / num_scales)
        output["rec_loss"] = sum(photo_losses) / num_scales
        output.update(losses)
        return output
