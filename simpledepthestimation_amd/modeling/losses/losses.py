"""SILog / variance losses (reference: detectron2/modeling/losses/losses.py:L5-18) on the HIP path."""
import torch.nn as nn

from ...hip import photometric as HP


class silog_loss(nn.Module):
    """forward(depth_est, depth_gt): masked (gt > 1) scale-invariant log loss x10.

    depth_gt may be at a HIGHER resolution than depth_est: it is then nearest-sampled inside the kernel, which is what
    Supervised.py:L44-45 does with resize_img(..., mode='nearest') before calling the loss.
    """

    def __init__(self, variance_focus):
        super().__init__()
        self.variance_focus = variance_focus

    def forward(self, depth_est, depth_gt):
        return HP.silog_loss(depth_est, depth_gt, self.variance_focus)

    def multi_scale(self, depth_ests, depth_gt, weights):
        """sum_k weights[k] * forward(depth_ests[k], depth_gt): the per-scale loop of Supervised.py:L42-47 as one launch per phase."""
        if len(depth_ests) > HP.MAX_SILOG_SCALES:
            terms = [self.forward(e, depth_gt) * w for e, w in zip(depth_ests, weights)]
            return sum(terms[1:], terms[0])
        return HP.silog_loss_multi(depth_ests, depth_gt, self.variance_focus, weights)


def variance_loss(depth):
    """losses.py:L16-18: 1 / mean((depth / mean(depth) - 1)^2) of one [B,1,h,w] fp32 depth map.  Two scalar reductions over a
    single-channel map: a handful of device-side torch reductions per scale (PackNet config only, weight 1e-4), no custom kernel."""
    return 1 / ((depth / depth.mean() - 1.0) ** 2).mean()
