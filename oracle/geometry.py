"""Oracle (test infrastructure): camera geometry and inverse warp, torch-CPU fp32.

Restates detectron2/geometry/camera.py and detectron2/geometry/pose_utils.py of the
reference.  The integer sample indices floor(ix), floor(iy) produced here are
bit-exact with the reference on the golden inputs (tests/test_oracle_golden.py);
to get there every 3-term dot product is an explicit ``p0*a0 -> fma -> fma`` chain,
which is what torch-CPU ``bmm`` does for K=3 (hard parts).
"""
import math

import torch

_FLT_MAX = 3.4028234663852886e38


def fma32(a, b, c):
    """fp32 fused multiply-add a*b+c with one rounding.

    a*b is exact in fp64 (24+24 significant bits), so fp64 add + one cast to fp32
    reproduces fmaf except for double-rounding ties (probability ~2^-29 per op).
    """
    return (a.double() * b.double() + c.double()).float()


def dot3_fma(a0, a1, a2, p0, p1, p2):
    """a0*p0 + a1*p1 + a2*p2 evaluated as mul, fma, fma (torch-CPU bmm order, K=3)."""
    acc = a0 * p0
    acc = fma32(a1, p1, acc)
    acc = fma32(a2, p2, acc)
    return acc


def dot3_plain(a0, a1, a2, p0, p1, p2):
    """(a0*p0 + a1*p1) + a2*p2 with every product and sum rounded (no FMA).

    This is what torch-CPU ``bmm`` does for [B,3,3] x [B,3,3]: below 400 multiply-adds per
    matrix it takes ATen's naive loop instead of MKL (measured bit-for-bit on 36k elements).
    """
    return (a0 * p0 + a1 * p1) + a2 * p2


def scale_intrinsics(K, x_scale, y_scale):
    """camera.py:L14-22 -- fx,cx *= x_scale ; fy,cy *= y_scale (returns a new tensor)."""
    K = K.clone()
    K[..., 0, 0] = K[..., 0, 0] * x_scale
    K[..., 1, 1] = K[..., 1, 1] * y_scale
    K[..., 0, 2] = K[..., 0, 2] * x_scale
    K[..., 1, 2] = K[..., 1, 2] * y_scale
    return K


def inv_intrinsics(K):
    """camera.py:L25-37 -- analytic inverse of an upper-triangular pinhole matrix."""
    fx, fy, cx, cy = K[:, 0, 0], K[:, 1, 1], K[:, 0, 2], K[:, 1, 2]
    Ki = K.clone()
    Ki[:, 0, 0] = 1.0 / fx
    Ki[:, 1, 1] = 1.0 / fy
    Ki[:, 0, 2] = -1.0 * cx / fx
    Ki[:, 1, 2] = -1.0 * cy / fy
    return Ki


def resize_bilinear_ac(img, size):
    """camera.py:L40-46 with mode='bilinear' (align_corners=True).

    Explicit restatement of torch's upsample_bilinear2d(align_corners=True):
    src = dst * (in-1)/(out-1); lambda1 = src - floor(src); lambda0 = 1 - lambda1;
    out = l0h*(l0w*v00 + l1w*v01) + l1h*(l0w*v10 + l1w*v11).
    """
    B, C, H, W = img.shape
    h, w = int(size[0]), int(size[1])
    if (H, W) == (h, w):
        return img
    sh = (H - 1) / (h - 1) if h > 1 else 0.0
    sw = (W - 1) / (w - 1) if w > 1 else 0.0
    ys = torch.arange(h, dtype=torch.float32) * torch.tensor(sh, dtype=torch.float32)
    xs = torch.arange(w, dtype=torch.float32) * torch.tensor(sw, dtype=torch.float32)
    y0 = ys.floor().long().clamp(max=H - 1)
    x0 = xs.floor().long().clamp(max=W - 1)
    y1 = (y0 + 1).clamp(max=H - 1)
    x1 = (x0 + 1).clamp(max=W - 1)
    ly1 = (ys - y0.float()).view(1, 1, h, 1)
    lx1 = (xs - x0.float()).view(1, 1, 1, w)
    ly0, lx0 = 1.0 - ly1, 1.0 - lx1
    r0 = img[:, :, y0, :]
    r1 = img[:, :, y1, :]
    top = lx0 * r0[:, :, :, x0] + lx1 * r0[:, :, :, x1]
    bot = lx0 * r1[:, :, :, x0] + lx1 * r1[:, :, :, x1]
    return ly0 * top + ly1 * bot


def resize_nearest(img, size):
    """camera.py:L40-46 with mode='nearest': src = floor(dst * in/out) (float scale)."""
    B, C, H, W = img.shape
    h, w = int(size[0]), int(size[1])
    if (H, W) == (h, w):
        return img
    ys = (torch.arange(h, dtype=torch.float32) * torch.tensor(H / h, dtype=torch.float32)).floor().long().clamp(max=H - 1)
    xs = (torch.arange(w, dtype=torch.float32) * torch.tensor(W / w, dtype=torch.float32)).floor().long().clamp(max=W - 1)
    return img[:, :, ys, :][:, :, :, xs]


def resize_img(img, size, mode="bilinear"):
    return resize_bilinear_ac(img, size) if mode == "bilinear" else resize_nearest(img, size)


def euler2mat(angle):
    """pose_utils.py:L98-127 -- R = X(rx) @ Y(ry) @ Z(rz), angle = (rx, ry, rz)."""
    x, y, z = angle[:, 0], angle[:, 1], angle[:, 2]
    zero = z.detach() * 0
    one = zero + 1
    cz, sz = torch.cos(z), torch.sin(z)
    cy, sy = torch.cos(y), torch.sin(y)
    cx, sx = torch.cos(x), torch.sin(x)
    zm = torch.stack([cz, -sz, zero, sz, cz, zero, zero, zero, one], 1).view(-1, 3, 3)
    ym = torch.stack([cy, zero, sy, zero, one, zero, -sy, zero, cy], 1).view(-1, 3, 3)
    xm = torch.stack([one, zero, zero, zero, cx, -sx, zero, sx, cx], 1).view(-1, 3, 3)
    return xm.bmm(ym).bmm(zm)


def pose_vec2mat(vec):
    """pose_utils.py:L130-137 -- vec = (tx,ty,tz,rx,ry,rz) -> [B,4,4] = [R t; 0 0 0 1]."""
    B = vec.shape[0]
    R = euler2mat(vec[:, 3:])
    top = torch.cat([R, vec[:, :3].unsqueeze(-1)], 2)
    bottom = torch.tensor([0.0, 0.0, 0.0, 1.0], dtype=vec.dtype).view(1, 1, 4).expand(B, 1, 4)
    return torch.cat([top, bottom], 1)


def warp_coords(depth, K, R, t):
    """Projection chain of view_synthesis (camera.py:L125-191), explicit fp32 op order.

    depth [B,1,H,W]; K [B,3,3]; R [B,3,3]; t [B,3] (constant per sample -- the
    intended MonoDepth2 semantics).
    Returns X, Y (un-normalised, before nan_to_num/clamp), Z (unclamped), all [B,H,W].
    """
    B, _, H, W = depth.shape
    Ki = inv_intrinsics(K)
    xs = torch.arange(W, dtype=torch.float32).view(1, 1, W).expand(B, H, W)
    ys = torch.arange(H, dtype=torch.float32).view(1, H, 1).expand(B, H, W)
    d = depth[:, 0]
    g0, g1, g2 = xs * d, ys * d, d                      # grid * depth           (L133)

    def e(M, i, j):
        return M[:, i, j].view(B, 1, 1)

    # points_A = Kinv.bmm(grid) + 0                                             (L136,L173)
    p = [dot3_fma(e(Ki, i, 0), e(Ki, i, 1), e(Ki, i, 2), g0, g1, g2) + 0.0 for i in range(3)]
    # R' = K.bmm(R) (3x3 @ 3x3: naive no-FMA path), t' = K.bmm(t) ([3,3] @ [3,HW]: MKL, FMA)  (L178-179)
    KR = [[dot3_plain(K[:, i, 0], K[:, i, 1], K[:, i, 2], R[:, 0, j], R[:, 1, j], R[:, 2, j]).view(B, 1, 1)
           for j in range(3)] for i in range(3)]
    Kt = [dot3_fma(K[:, i, 0], K[:, i, 1], K[:, i, 2], t[:, 0], t[:, 1], t[:, 2]).view(B, 1, 1) for i in range(3)]
    # proj = R'.bmm(points) + t'                                                (L147)
    q = [dot3_fma(KR[i][0], KR[i][1], KR[i][2], p[0], p[1], p[2]) + Kt[i] for i in range(3)]
    X = q[0] / (q[2] + 1e-6)
    Y = q[1] / (q[2] + 1e-6)
    return X, Y, q[2]


def grid_sample_bilinear_zeros_ac(img, ix, iy):
    """F.grid_sample(bilinear, zeros, align_corners=True) given UN-normalised coords.

    img [B,C,H,W]; ix, iy [B,h,w] pixel coordinates.  Weights follow ATen's CPU kernel:
    w = ix - floor(ix), e = 1 - w, n = iy - floor(iy), s = 1 - n;
    out = v_nw*(s*e) + v_ne*(s*w) + v_sw*(n*e) + v_se*(n*w); out-of-range taps read 0.
    Differentiable w.r.t. ix, iy (and img).
    """
    B, C, H, W = img.shape
    x0f, y0f = ix.detach().floor(), iy.detach().floor()
    wx = ix - x0f
    ex = 1.0 - wx
    ny = iy - y0f
    sy = 1.0 - ny
    x0, y0 = x0f.long(), y0f.long()
    flat = img.reshape(B, C, H * W)

    def tap(yy, xx):
        ok = ((xx >= 0) & (xx < W) & (yy >= 0) & (yy < H))
        idx = (yy.clamp(0, H - 1) * W + xx.clamp(0, W - 1)).view(B, 1, -1).expand(B, C, -1)
        v = flat.gather(2, idx).view(B, C, *ix.shape[1:])
        return v * ok.unsqueeze(1).to(v.dtype)

    out = tap(y0, x0) * (sy * ex).unsqueeze(1)
    out = out + tap(y0, x0 + 1) * (sy * wx).unsqueeze(1)
    out = out + tap(y0 + 1, x0) * (ny * ex).unsqueeze(1)
    out = out + tap(y0 + 1, x0 + 1) * (ny * wx).unsqueeze(1)
    return out


def view_synthesis(image_B, depth_A, K, R, t):
    """camera.py:L166-202.  t is [B,3] (per-sample constant).

    Returns dict: sampled [B,C,H,W], Z [B,1,H,W] (clamped 1e-5), grid [B,H,W,2] (normalised),
    valid [B,1,H,W] bool, fx/fy = floor of the un-normalised sample coordinate (int32),
    ix/iy = un-normalised sample coordinate.
    """
    B, _, H, W = depth_A.shape
    X, Y, Z = warp_coords(depth_A, K, R, t)
    valid = (torch.isfinite(X) & (X >= 0) & (X < W - 1) & torch.isfinite(Y) & (Y >= 0) & (Y < H - 1) & (Z > 0))
    Zc = Z.clamp(min=1e-5)
    Xs = torch.nan_to_num(X).clamp(0, W - 1)
    Ys = torch.nan_to_num(Y).clamp(0, H - 1)
    xn = 2 * Xs / (W - 1) - 1.0
    yn = 2 * Ys / (H - 1) - 1.0
    # ATen CPU un-normalise (align_corners=True): (coord + 1) * ((size - 1) / 2)
    ix = (xn + 1.0) * ((W - 1) / 2)
    iy = (yn + 1.0) * ((H - 1) / 2)
    sampled = grid_sample_bilinear_zeros_ac(image_B, ix, iy)
    return {
        "sampled": sampled,
        "Z": Zc.unsqueeze(1),
        "grid": torch.stack([xn, yn], -1),
        "valid": valid.unsqueeze(1),
        "fx": ix.detach().floor().to(torch.int32),
        "fy": iy.detach().floor().to(torch.int32),
        "ix": ix,
        "iy": iy,
    }
