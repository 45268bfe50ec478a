"""SupDepthModel on the HIP path.  Contract of detectron2/modeling/meta_arch/Supervised.py:L18-49: training returns the batch dict with
``silog_loss`` = mean over the four scales of SILog(pred_s, nearest-resized GT); eval returns ``depth_pred`` = the full-resolution map."""
from ..losses.losses import silog_loss
from .build import META_ARCH_REGISTRY
from .common import HipMetaArch


@META_ARCH_REGISTRY.register()
class SupDepthModel(HipMetaArch):
    def __init__(self, cfg):
        HipMetaArch.__init__(self, cfg)
        self.loss = silog_loss(cfg.LOSS.VARIANCE_FOCUS)

    def forward(self, batch):
        result = self.run_depth_net(batch)
        scales = result["depth_pred"]
        if not self.training:
            result["depth_pred"] = scales[0]
            return result
        # the nearest resize of the ground truth to each scale (Supervised.py:L44) happens inside the loss kernel
        result["silog_loss"] = self.loss.multi_scale(list(scales), result["depth"], [1.0 / len(scales)] * len(scales))
        return result
