"""SSIM lives inside the fused photometric kernel (sde_photo_fwd); this module only carries its constants so that
``MonoDepth2Model.ssim`` exists as in the reference (detectron2/modeling/losses/ssim_loss.py:L6-53)."""
import torch.nn as nn


class SSIM(nn.Module):
    def __init__(self, C1=1e-4, C2=9e-4, kernel_size=3, stride=1):
        super().__init__()
        if kernel_size != 3 or stride != 1:
            raise NotImplementedError("the fused kernel implements the 3x3 / stride-1 SSIM the reference uses")
        self.C1, self.C2 = C1, C2

    def forward(self, x, y):
        raise RuntimeError("SSIM is evaluated inside hip.photometric.photometric_scale_loss; it has no stand-alone entry point")
