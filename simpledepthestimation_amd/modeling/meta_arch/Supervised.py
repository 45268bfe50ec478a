"""SupDepthModel (reference: detectron2/modeling/meta_arch/Supervised.py:L18-49)."""
import torch
import torch.nn as nn

from ...hip import nn as HN
from ...utils.memory import to_cuda
from ..depth_net import build_depth_net
from ..losses.losses import silog_loss
from .build import META_ARCH_REGISTRY


@META_ARCH_REGISTRY.register()
class SupDepthModel(nn.Module):
    def __init__(self, cfg):
        super().__init__()
        self.depth_net = build_depth_net(cfg)
        self.loss = silog_loss(cfg.LOSS.VARIANCE_FOCUS)
        self.register_buffer("pixel_mean", torch.Tensor(cfg.MODEL.PIXEL_MEAN).view(1, -1, 1, 1))
        self.register_buffer("pixel_std", torch.Tensor(cfg.MODEL.PIXEL_STD).view(1, -1, 1, 1))

    @property
    def device(self):
        return self.pixel_mean.device

    def forward(self, batch):
        batch = to_cuda(batch, self.device)
        # (img - mean)/std, NCHW->NHWC, channel padding, dtype cast and the optional flip: one kernel
        batch["depth_net_input_nhwc"] = HN.prep_input(batch["img"], self.pixel_mean, self.pixel_std, self.depth_net.dtype,
                                                      bool(batch.get("flip", False)))
        output = self.depth_net(batch)
        if self.training:
            # resize_img(depth, pred.shape, 'nearest') is folded into the loss kernel (Supervised.py:L44-45)
            sup_losses = [self.loss(pred, batch["depth"]) for pred in output["depth_pred"]]
            output["silog_loss"] = sum(sup_losses) / len(sup_losses)
        else:
            output["depth_pred"] = output["depth_pred"][0]
        return output
