"""SSIM distance module with the reference's interface (detectron2/modeling/losses/ssim_loss.py:L6-53).  The training path evaluates SSIM
inside the fused photometric kernels (sde_photo_fwd / sde_photo_bwd, which read ``C1`` / ``C2`` from this module); calling the module itself
returns the stand-alone map through sde_ssim_fwd, differentiable in both images (sde_ssim_bwd)."""
import torch.nn as nn

from ...hip import photometric as HP


class SSIM(nn.Module):
    def __init__(self, C1=1e-4, C2=9e-4, kernel_size=3, stride=1):
        super().__init__()
        if kernel_size != 3 or stride != 1:
            raise NotImplementedError("the HIP kernels implement the 3x3 / stride-1 SSIM the reference uses")
        self.C1, self.C2 = C1, C2

    def forward(self, x, y):
        """x, y: [B,C,H,W] -> clamp((1 - SSIM) / 2, 0, 1) per pixel and channel (ReflectionPad2d(1) + 3x3 mean)."""
        return HP.ssim_map(x, y, self.C1, self.C2)
