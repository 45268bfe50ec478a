"""What the two meta-architectures share on the HIP path: the normalisation constants as buffers (state-dict keys ``pixel_mean`` /
``pixel_std``, as in the reference), the ``device`` property the reference's loops read, and the fused input preparation."""
import torch
from torch import nn

from ...hip import nn as HN
from ...utils.memory import to_cuda
from ..depth_net import build_depth_net


class HipMetaArch(nn.Module):
    def __init__(self, cfg):
        nn.Module.__init__(self)
        for name, values in (("pixel_mean", cfg.MODEL.PIXEL_MEAN), ("pixel_std", cfg.MODEL.PIXEL_STD)):
            self.register_buffer(name, torch.tensor(list(values), dtype=torch.float32).reshape(1, len(values), 1, 1))
        self.depth_net = build_depth_net(cfg)

    @property
    def device(self):
        return self.pixel_mean.device

    def run_depth_net(self, batch):
        """Moves the batch to the model's device and runs the depth network on the fused NHWC input: (img - mean) / std, NCHW -> NHWC,
        channel padding, dtype cast and the optional horizontal flip are ONE kernel (sde_prep_input)."""
        batch = to_cuda(batch, self.device)
        batch["depth_net_input_nhwc"] = HN.prep_input(batch["img"], self.pixel_mean, self.pixel_std, self.depth_net.dtype, bool(batch.get("flip", False)))
        return self.depth_net(batch)
