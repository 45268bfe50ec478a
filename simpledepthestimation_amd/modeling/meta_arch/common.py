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

    def _weighted_sum(self, values, weights):
        """sum_i weights[i] * values[i] of 0-d loss tensors as one stack + one dot product with a cached device vector.  The reference accumulates
        `loss += term_i * w_i` scale by scale: two tiny kernels per scale and loss forward, as many backward."""
        v0 = values if torch.is_tensor(values) else values[0]
        key = (tuple(float(w) for w in weights), v0.device, v0.dtype)
        cache = self.__dict__.setdefault("_wsum_cache", {})
        if key not in cache:      # built during the first (eager) step, before any hipGraph capture: no host-to-device copy inside a captured step
            cache[key] = torch.tensor(key[0], device=key[1], dtype=key[2])
        if torch.is_tensor(values):       # already one [n] tensor (the multi-scale photometric launch): the dot product alone
            return torch.dot(values, cache[key])
        return torch.dot(torch.stack([v.reshape(()) for v in values]), cache[key])

    def run_depth_net(self, batch):
        """Moves the batch to the model's device and runs the depth network on the fused NHWC input: (img - mean) / std, NCHW -> NHWC,
        channel padding, dtype cast and the optional horizontal flip are ONE kernel (sde_prep_input)."""
        batch = to_cuda(batch, self.device)
        batch["depth_net_input_nhwc"] = HN.prep_input(batch["img"], self.pixel_mean, self.pixel_std, self.depth_net.dtype, bool(batch.get("flip", False)))
        return self.depth_net(batch)
