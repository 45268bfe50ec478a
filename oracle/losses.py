"""Oracle (test infrastructure): SSIM / photometric / smoothness / SILog, torch-CPU fp32.

Restates detectron2/modeling/losses/{ssim_loss,smoothness_loss,losses}.py and
MonoDepth2Model.rgb_consistency_loss (meta_arch/MonoDepth2.py:L130-151).
"""
import torch
import torch.nn.functional as F

from . import geometry as G


def _box3_reflect(x):
    """ReflectionPad2d(1) then 3x3 mean, stride 1 (ssim_loss.py:L31-37)."""
    return F.avg_pool2d(F.pad(x, (1, 1, 1, 1), mode="reflect"), 3, 1)


def ssim_distance(x, y, C1=1e-4, C2=9e-4):
    """ssim_loss.py:L34-53 -- clamp((1 - SSIM)/2, 0, 1) per pixel and channel."""
    mu_x, mu_y = _box3_reflect(x), _box3_reflect(y)
    mu_xy, mu_xx, mu_yy = mu_x * mu_y, mu_x.pow(2), mu_y.pow(2)
    s_x = _box3_reflect(x.pow(2)) - mu_xx
    s_y = _box3_reflect(y.pow(2)) - mu_yy
    s_xy = _box3_reflect(x * y) - mu_xy
    n = (2 * mu_xy + C1) * (2 * s_xy + C2)
    d = (mu_xx + mu_yy + C1) * (s_x + s_y + C2)
    return torch.clamp((1.0 - n / d) / 2.0, 0.0, 1.0)


def photometric_map(sampled_B, frame_A, ssim_w=0.85, C1=1e-4, C2=9e-4, clip=0.0):
    """MonoDepth2.py:L137-151 -- ssim_w*mean_c(SSIM) + (1-ssim_w)*mean_c|B-A|, [B,1,h,w]."""
    l1 = (sampled_B - frame_A).abs().mean(1, True)
    if ssim_w > 0.0:
        s = ssim_distance(sampled_B, frame_A, C1, C2).mean(1, True)
        l1 = s * ssim_w + l1 * (1 - ssim_w)
    if clip > 0.0:
        m, sd = l1.mean(), l1.std()
        l1 = torch.clamp(l1, max=float(m + clip * sd))
    return l1


def rgb_consistency(frame_A, frame_B, depth_A, K, R=None, t=None, **kw):
    """MonoDepth2.py:L130-151 with the intended [B,3] translation."""
    if R is not None and t is not None:
        sampled = G.view_synthesis(frame_B, depth_A, K, R, t)["sampled"]
    else:
        sampled = frame_B
    return photometric_map(sampled, frame_A, **kw)


def min_reprojection(maps):
    """MonoDepth2.py:L119 -- cat(maps,1).min(1).mean()."""
    return torch.cat(maps, 1).min(1, True)[0].mean()


def smoothness(depth, image):
    """smoothness_loss.py:L42-80 (reversed=False)."""
    inv = 1.0 / depth.clamp(min=1e-6)
    mean_inv = inv.mean(2, True).mean(3, True)
    nrm = inv / mean_inv.clamp(min=1e-6)
    dgx = nrm[:, :, :, :-1] - nrm[:, :, :, 1:]
    dgy = nrm[:, :, :-1, :] - nrm[:, :, 1:, :]
    igx = image[:, :, :, :-1] - image[:, :, :, 1:]
    igy = image[:, :, :-1, :] - image[:, :, 1:, :]
    wx = torch.exp(-igx.abs().mean(1, True))
    wy = torch.exp(-igy.abs().mean(1, True))
    return (dgx * wx).abs().mean() + (dgy * wy).abs().mean()


def silog(depth_est, depth_gt, variance_focus=0.85):
    """losses.py:L10-13 -- masked (gt > 1) scale-invariant log loss, x10."""
    mask = depth_gt > 1.0
    d = torch.log(depth_est[mask]) - torch.log(depth_gt[mask])
    return torch.sqrt((d ** 2).mean() - variance_focus * (d.mean() ** 2)) * 10.0


def variance(depth):
    """losses.py:L16-18."""
    return 1 / ((depth / depth.mean() - 1.0) ** 2).mean()
