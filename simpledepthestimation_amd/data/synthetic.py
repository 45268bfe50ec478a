"""Synthetic KITTI-shaped batches: what bench.py, the smoke test and the parity tests feed the models.

Deterministic across hosts (torch CPU generator).  The golden fixtures under tests/golden were generated from inputs drawn by exactly
these functions (oracle/gen_golden.py imports them), so they must not change arithmetic."""
import torch


def kitti_K(B, H, W):
    """KITTI-normalised intrinsics (fx = 0.58 W, fy = 1.92 H, principal point at the centre), [B,3,3]."""
    return torch.tensor([[0.58 * W, 0, 0.5 * W], [0, 1.92 * H, 0.5 * H], [0, 0, 1.0]]).repeat(B, 1, 1)


def smooth_images(g, B, H, W, n=3):
    """Low-pass noise images in [0,1] (so the warp / SSIM see structure, not white noise)."""
    out = []
    for _ in range(n):
        x = torch.rand(B, 3, H // 4 + 2, W // 4 + 2, generator=g)
        x = torch.nn.functional.interpolate(x, size=(H, W), mode="bicubic", align_corners=False).clamp(0, 1)
        out.append(x.contiguous())
    return out


def sup_batch(B, H, W, seed):
    """Supervised batch: img U[0,1), depth = 30 % dense U[1,80) else 0 (no ground truth)."""
    g = torch.Generator().manual_seed(seed)
    img = torch.rand(B, 3, H, W, generator=g)
    m = torch.rand(B, 1, H, W, generator=g) < 0.3
    depth = torch.where(m, torch.rand(B, 1, H, W, generator=g) * 79 + 1, torch.zeros(1))
    return {"img": img, "depth": depth}


def mono_batch(B, H, W, seed):
    """MonoDepth2 batch: target + two context frames (smooth images), KITTI intrinsics."""
    g = torch.Generator().manual_seed(seed)
    a, b, c = smooth_images(g, B, H, W)
    return {"img": a, "img_orig": a.clone(), "ctx_img": [b, c], "ctx_img_orig": [b.clone(), c.clone()],
            "intrinsics": kitti_K(B, H, W)}
