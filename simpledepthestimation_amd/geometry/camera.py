"""Camera geometry entry points with the reference's names (detectron2/geometry/camera.py), computed by libsde_hip.so."""
import torch

from ..hip import photometric as HP


def scale_intrinsics(K, x_scale, y_scale):
    """camera.py:L14-22 (in place, like the reference).  The fused loss kernels take the scale factors instead."""
    K[..., 0, 0] *= x_scale
    K[..., 1, 1] *= y_scale
    K[..., 0, 2] *= x_scale
    K[..., 1, 2] *= y_scale
    return K


def resize_img(image, dst_size, mode="bilinear"):
    """camera.py:L40-46."""
    return HP.resize(image, dst_size, mode)


def view_synthesis(image_B, depth_A, intrinsics, R_A_to_B, t_A_to_B):
    """camera.py:L166-202 -> (sampled_B, depth_in_B, grid, valid_mask).  t may be [B,3,1,1], [B,3,H,W] (constant) or [B,3]."""
    B = image_B.shape[0]
    t = t_A_to_B.reshape(B, 3, -1)[:, :, 0]
    pose = torch.zeros(B, 4, 4, device=image_B.device, dtype=torch.float32)
    pose[:, :3, :3] = R_A_to_B
    pose[:, :3, 3] = t
    pose[:, 3, 3] = 1.0
    o = HP.view_synthesis_raw(image_B, depth_A, intrinsics, pose, 1.0, 1.0, want_indices=False)
    return o["sampled"], o["Z"], o["grid"], o["valid"].bool()
