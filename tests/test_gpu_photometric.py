"""GPU parity: photometric path of libsde_hip.so (through the C ABI) vs the CPU oracle and the reference goldens.

Tolerances (BASELINE.json north_star): warp sample indices bit-exact; fp32 values 1e-5 abs/rel unless noted.
"""
import numpy as np
import pytest
import torch

from oracle import geometry as G, losses as OL
from oracle.gen_golden import POSE_VECS, kitti_K, smooth_images

pytestmark = pytest.mark.gpu
dev = "cuda"


@pytest.fixture(scope="module")
def P():
    from simpledepthestimation_amd.hip import photometric
    return photometric


def close(a, b, rtol=1e-5, atol=1e-6, frac=0.0):
    a = torch.as_tensor(a).detach().cpu().double(); b = torch.as_tensor(b).detach().cpu().double()
    assert a.shape == b.shape, (a.shape, b.shape)
    bad = ((a - b).abs() > atol + rtol * b.abs())
    nbad = int(bad.sum())
    assert nbad <= frac * a.numel(), f"{nbad}/{a.numel()} out of tolerance, max abs err {(a - b).abs().max().item():.3e}"


@pytest.mark.parametrize("tag", ["s48", "s24", "hard"])
def test_view_synthesis_golden(P, geo, tag):
    Bf, D, K, Pm = (geo.t(f"{tag}.{k}") for k in ("Bf", "D", "K", "P"))
    out = P.view_synthesis_raw(Bf.to(dev), D.to(dev), K.to(dev), Pm.to(dev))
    assert torch.equal(out["fx"].cpu(), geo.t(f"{tag}.fx")), "floor(ix) not bit-exact vs reference golden"
    assert torch.equal(out["fy"].cpu(), geo.t(f"{tag}.fy")), "floor(iy) not bit-exact vs reference golden"
    assert torch.equal(out["grid"].cpu(), geo.t(f"{tag}.grid")), "normalised grid not bit-exact"
    assert torch.equal(out["valid"].cpu().bool(), geo.t(f"{tag}.valid"))
    close(out["Z"], geo.t(f"{tag}.Z"), 1e-6, 0)
    close(out["sampled"], geo.t(f"{tag}.sampled"), 1e-5, 2e-6)


def test_view_synthesis_full_size_bit_exact(P, geo):
    """BASELINE size (192x640, here B=12): indices bit-exact vs the oracle + reference known answers for B=2."""
    torch.manual_seed(1234)
    B, H, W = 2, 192, 640
    A = torch.rand(B, 3, H, W); Bf = torch.rand(B, 3, H, W); D = torch.rand(B, 1, H, W) * 79 + 1
    Pm = geo.t("full.P")
    out = P.view_synthesis_raw(Bf.to(dev), D.to(dev), kitti_K(B, H, W).to(dev), Pm.to(dev))
    assert int(out["fx"].long().sum()) == 78389238 and int(out["fy"].long().sum()) == 23330106   # SURVEY.md 8c literals
    assert torch.equal(out["fx"].cpu()[:, ::8, ::8], geo.t("full.fx_sub"))
    g = torch.Generator().manual_seed(5)
    B = 12
    Bf = torch.rand(B, 3, H, W, generator=g); D = torch.rand(B, 1, H, W, generator=g) * 60 + 0.5
    vec = (torch.rand(B, 6, generator=g) - 0.5) * torch.tensor([0.6, 0.2, 1.0, 0.02, 0.04, 0.02])
    Pm = G.pose_vec2mat(vec)
    K = kitti_K(B, H, W)
    ref = G.view_synthesis(Bf, D, K, Pm[:, :3, :3], Pm[:, :3, 3])
    out = P.view_synthesis_raw(Bf.to(dev), D.to(dev), K.to(dev), Pm.to(dev))
    assert torch.equal(out["fx"].cpu(), ref["fx"]) and torch.equal(out["fy"].cpu(), ref["fy"])
    assert torch.equal(out["valid"].cpu().bool(), ref["valid"])
    close(out["sampled"], ref["sampled"], 1e-5, 2e-6)


def test_scaled_intrinsics_path(P):
    """sx, sy != 1: the kernel applies scale_intrinsics itself (camera.py:L14-22)."""
    g = torch.Generator().manual_seed(2)
    B, H, W = 3, 48, 160
    Bf = torch.rand(B, 3, H, W, generator=g); D = torch.rand(B, 1, H, W, generator=g) * 40 + 1
    Pm = G.pose_vec2mat(torch.cat([POSE_VECS, POSE_VECS[:1] * 0.5], 0))
    Kfull = kitti_K(B, 192, 640)
    Ks = G.scale_intrinsics(Kfull, 0.25, 0.25)
    ref = G.view_synthesis(Bf, D, Ks, Pm[:, :3, :3], Pm[:, :3, 3])
    out = P.view_synthesis_raw(Bf.to(dev), D.to(dev), Kfull.to(dev), Pm.to(dev), 0.25, 0.25)
    assert torch.equal(out["fx"].cpu(), ref["fx"]) and torch.equal(out["fy"].cpu(), ref["fy"])


def test_identity_pose_is_identity(P):
    """Property: R=I, t=0 => the warp reproduces the image on the interior."""
    g = torch.Generator().manual_seed(3)
    B, H, W = 2, 96, 320
    img = torch.rand(B, 3, H, W, generator=g); D = torch.rand(B, 1, H, W, generator=g) * 30 + 2
    eye = torch.eye(4).repeat(B, 1, 1)
    out = P.view_synthesis_raw(img.to(dev), D.to(dev), kitti_K(B, H, W).to(dev), eye.to(dev))
    close(out["sampled"][:, :, 1:-1, 1:-1], img[:, :, 1:-1, 1:-1], 1e-4, 1e-4, frac=1e-3)


def test_resize(P, geo):
    img, dep = geo.t("resize.img"), geo.t("resize.dep")
    for h, w in [(24, 80), (12, 40), (6, 20)]:
        close(P.resize(img.to(dev), (h, w)), geo.t(f"resize.bil_{h}"), 1e-6, 1e-6)
        assert torch.equal(P.resize(dep.to(dev), (h, w), mode="nearest").cpu(), geo.t(f"resize.nn_{h}"))
    g = torch.Generator().manual_seed(4)
    big = torch.rand(12, 3, 192, 640, generator=g)
    for h, w in [(96, 320), (48, 160), (24, 80)]:
        close(P.resize(big.to(dev), (h, w)), G.resize_img(big, (h, w)), 1e-6, 1e-6)
    assert P.resize(big.to(dev), (192, 640)).shape == big.shape


def test_pose_vec2mat(P, geo):
    vec = geo.t("misc.vec")
    close(P.pose_vec2mat(vec.to(dev)), geo.t("misc.P"), 1e-6, 1e-7)
    v1 = vec.clone().requires_grad_(True); v2 = vec.clone().to(dev).requires_grad_(True)
    w = torch.randn(vec.shape[0], 4, 4, generator=torch.Generator().manual_seed(1))
    (G.pose_vec2mat(v1) * w).sum().backward()
    (P.pose_vec2mat(v2) * w.to(dev)).sum().backward()
    close(v2.grad, v1.grad, 1e-5, 1e-6)


def _photo_case(B, h, w, seed, scale=1.0):
    g = torch.Generator().manual_seed(seed)
    A, C0, C1 = smooth_images(g, B, h, w)
    D = torch.rand(B, 1, h, w, generator=g) * 30 + 2
    # smooth the depth a little so that bilinear gradients are informative
    D = torch.nn.functional.avg_pool2d(torch.nn.functional.pad(D, (2, 2, 2, 2), mode="replicate"), 5, 1)
    vec = torch.tensor([[0.05, -0.01, 0.3, 0.002, -0.004, 0.001], [-0.03, 0.02, -0.25, -0.001, 0.003, 0.002],
                        [0.1, 0.0, -0.1, 0.0, 0.01, 0.0]])[:B]
    if B > 3:
        vec = torch.cat([vec, (torch.rand(B - 3, 6, generator=g) - 0.5) * 0.1], 0)
    Kfull = kitti_K(B, int(h / scale), int(w / scale))
    return A, [C0, C1], D, Kfull, [G.pose_vec2mat(vec), G.pose_vec2mat(-vec)]


def _oracle_photo(A, ctxs, D, Kfull, poses, sx, sy, automask=True, reduce="min", ssim_w=0.85):
    Ks = G.scale_intrinsics(Kfull, sx, sy)
    maps = []
    for c, p in zip(ctxs, poses):
        maps.append(OL.rgb_consistency(A, c, D, Ks, p[:, :3, :3], p[:, :3, 3], ssim_w=ssim_w))
        if automask:
            maps.append(OL.rgb_consistency(A, c, D, Ks, None, None, ssim_w=ssim_w))
    loss = OL.min_reprojection(maps) if reduce == "min" else sum(m.mean() for m in maps) / len(maps)
    return loss, maps


@pytest.mark.parametrize("B,h,w,scale", [(2, 48, 160, 0.25), (3, 24, 80, 0.125), (2, 64, 192, 1.0), (1, 37, 75, 1.0)])
@pytest.mark.parametrize("automask,reduce,ssim_w", [(True, "min", 0.85), (False, "min", 0.85), (True, "mean", 0.85), (True, "min", 0.0)])
def test_photometric_forward(P, B, h, w, scale, automask, reduce, ssim_w):
    A, ctxs, D, Kfull, poses = _photo_case(B, h, w, 10 + h, scale)
    loss_o, maps_o = _oracle_photo(A, ctxs, D, Kfull, poses, scale, scale, automask, reduce, ssim_w)
    out = P.photometric_maps(D.to(dev), Kfull.to(dev), A.to(dev), [c.to(dev) for c in ctxs], [p.to(dev) for p in poses], scale, scale,
                             ssim_w=ssim_w, automask=automask, reduce=reduce)
    close(out["maps"], torch.cat(maps_o, 1), 1e-4, 2e-5)
    close(out["loss"], loss_o, 2e-5, 1e-6)
    if reduce == "min":
        cat = torch.cat(maps_o, 1)
        mism = (out["sel"].cpu().long() != cat.argmin(1)).float().mean().item()
        assert mism < 2e-3, f"arg-min map mismatch fraction {mism}"      # only at fp32 near-ties


@pytest.mark.parametrize("B,h,w,scale", [(2, 48, 160, 0.25), (2, 24, 80, 0.125), (1, 37, 75, 1.0)])
@pytest.mark.parametrize("automask,reduce", [(True, "min"), (False, "min"), (True, "mean")])
def test_photometric_backward(P, B, h, w, scale, automask, reduce):
    A, ctxs, D, Kfull, poses = _photo_case(B, h, w, 20 + h, scale)
    Dg = D.clone().requires_grad_(True)
    pg = [p.clone().requires_grad_(True) for p in poses]
    loss_o, _ = _oracle_photo(A, ctxs, Dg, Kfull, pg, scale, scale, automask, reduce)
    (loss_o * 3.0).backward()
    Dd = D.clone().to(dev).requires_grad_(True)
    pd = [p.clone().to(dev).requires_grad_(True) for p in poses]
    loss = P.photometric_scale_loss(Dd, Kfull.to(dev), A.to(dev), [c.to(dev) for c in ctxs], pd, scale, scale, automask=automask, reduce=reduce)
    (loss * 3.0).backward()
    close(loss, loss_o, 2e-5, 1e-6)
    gd, go = Dd.grad.cpu().double(), Dg.grad.double()
    rel = (gd - go).norm() / go.norm()
    assert rel < 2e-3, f"d_depth relative L2 error {rel:.3e}"
    close(gd, go, 1e-3, 1e-3 * go.abs().max().item(), frac=2e-3)
    for j in range(2):
        gp, gq = pd[j].grad.cpu().double()[:, :3], pg[j].grad.double()[:, :3]
        rel = (gp - gq).norm() / gq.norm()
        assert rel < 5e-3, f"d_pose[{j}] relative L2 error {rel:.3e}"


@pytest.mark.parametrize("automask,reduce", [(True, "min"), (False, "min"), (True, "mean")])
@pytest.mark.parametrize("B,H,W,n", [(2, 64, 192, 4), (1, 40, 72, 3), (3, 32, 96, 2), (2, 16, 24, 1)])
def test_photometric_all_scales_in_one_launch(P, B, H, W, n, automask, reduce):
    """sde_photo_multi_fwd / _bwd (MonoDepth2.py:L78-112: the loop over the decoder scales as one launch per phase) against n calls of the single-scale
    entry points: per-scale losses and depth gradients bit-identical (same workgroup body, same partial order), pose gradients = their sum over the scales."""
    g = torch.Generator().manual_seed(B * 100 + n)
    scales, cases = [], []
    for i in range(n):
        h, w = H >> i, W >> i
        A, ctxs, D, Kfull, poses = _photo_case(B, h, w, 30 + i, scale=h / H)
        cases.append((A, ctxs, D))
        scales.append((w / W, h / H))
    _, _, _, Kfull, poses = _photo_case(B, H, W, 30, 1.0)
    K = Kfull.to(dev)
    wts = torch.rand(n, generator=g) + 0.5

    def run(multi):
        Ds = [c[2].clone().to(dev).requires_grad_(True) for c in cases]
        pd = [p.clone().to(dev).requires_grad_(True) for p in poses]
        As = [c[0].to(dev) for c in cases]
        Cs = [[x.to(dev) for x in c[1]] for c in cases]
        if multi:
            losses = P.photometric_multi_loss(Ds, K, As, Cs, pd, scales, automask=automask, reduce=reduce)
        else:
            losses = torch.stack([P.photometric_scale_loss(Ds[i], K, As[i], Cs[i], pd, scales[i][0], scales[i][1], automask=automask, reduce=reduce) for i in range(n)])
        (losses * wts.to(dev)).sum().backward()
        torch.cuda.synchronize()
        return losses.detach().cpu(), [d.grad.cpu() for d in Ds], [q.grad.cpu() for q in pd]
    l1, g1, p1 = run(False)
    lm, gm, pm = run(True)
    assert torch.equal(lm, l1), (lm, l1)
    for i in range(n):
        assert torch.equal(gm[i], g1[i]), f"d_depth of scale {i} differs"
    for j in range(2):
        close(pm[j][:, :3], p1[j][:, :3], 1e-5, 1e-6 * float(p1[j].abs().max()))
        assert (pm[j][:, 3] == 0).all()


@pytest.mark.parametrize("automask,reduce,smooth", [(True, "min", True), (False, "min", True), (True, "mean", True), (True, "min", False)])
@pytest.mark.parametrize("B,H,W,n", [(2, 64, 192, 4), (1, 40, 72, 3), (2, 16, 24, 1)])
def test_mono_loss_all_scales(P, B, H, W, n, automask, reduce, smooth):
    """sde_mono_loss_fwd / _bwd (MonoDepth2.py:L78-126: photometric + smoothness terms of every decoder scale, weighted and summed) against the per-scale,
    per-term entry points: every scale's two terms bit-identical, the weighted totals and all gradients equal up to fp32 re-association of the weights."""
    from simpledepthestimation_amd.modeling.losses.smoothness_loss import smoothness_loss
    g = torch.Generator().manual_seed(B * 100 + n + 7)
    scales, cases = [], []
    for i in range(n):
        h, w = H >> i, W >> i
        A, ctxs, D, Kfull, poses = _photo_case(B, h, w, 40 + i, scale=h / H)
        cases.append((A, ctxs, D))
        scales.append((w / W, h / H))
    _, _, _, Kfull, poses = _photo_case(B, H, W, 40, 1.0)
    K = Kfull.to(dev)
    pw = [float(v) for v in (torch.rand(n, generator=g) + 0.5)]
    sw = [float(v) for v in (torch.rand(n, generator=g) * 1e-2 + 1e-3)] if smooth else None
    up = torch.tensor([1.7, 0.6])

    def run(fused):
        Ds = [c[2].clone().to(dev).requires_grad_(True) for c in cases]
        pd = [p.clone().to(dev).requires_grad_(True) for p in poses]
        As = [c[0].to(dev) for c in cases]
        Cs = [[x.to(dev) for x in c[1]] for c in cases]
        if fused:
            rec, sm, per = P.mono_loss(Ds, K, As, Cs, pd, scales, pw, sw, automask=automask, reduce=reduce)
            per = per.detach().cpu()
        else:
            ph = [P.photometric_scale_loss(Ds[i], K, As[i], Cs[i], pd, scales[i][0], scales[i][1], automask=automask, reduce=reduce) for i in range(n)]
            ss = [smoothness_loss(Ds[i], As[i]) for i in range(n)] if smooth else []
            rec = sum(ph[i] * pw[i] for i in range(n))
            sm = sum(ss[i] * sw[i] for i in range(n)) if smooth else torch.zeros((), device=dev)
            per = torch.stack([x.detach() for x in ph + ss] + [torch.zeros((), device=dev)] * (0 if smooth else n)).cpu()
        tot = rec * up[0].item() + (sm * up[1].item() if smooth else 0.0)
        tot.backward()
        torch.cuda.synchronize()
        return float(rec), float(sm), per, [d.grad.cpu() for d in Ds], [q.grad.cpu() for q in pd]
    r1, s1, per1, g1, p1 = run(False)
    rm, smm, perm, gm, pm = run(True)
    k = 2 * n if smooth else n
    assert torch.equal(perm[:k], per1[:k]), (perm, per1)
    assert abs(rm - r1) <= 3e-7 * abs(r1) and abs(smm - s1) <= 3e-7 * abs(s1) + 1e-12
    for i in range(n):
        # (the upstream gradient enters the kernels as g * (w / N) here and (g * w) * (1 / N) there, before sums with cancellation)
        close(gm[i], g1[i], 1e-5, 2e-5 * float(g1[i].abs().max()))
    for j in range(2):
        close(pm[j][:, :3], p1[j][:, :3], 1e-5, 1e-6 * float(p1[j].abs().max()))
        assert (pm[j][:, 3] == 0).all()
    # the arrival counter of the finalize kernel is left at zero (every launch depends on it)
    assert int(P._ticket(K.device).item()) == 0


def test_photometric_full_size_properties(P):
    """BASELINE size B=12, 192x640: size-independent properties (too slow for the oracle's autograd in CI seconds)."""
    B, h, w = 12, 192, 640
    A, ctxs, D, Kfull, poses = _photo_case(B, h, w, 99)
    Ad, Cd, Dd, Kd = A.to(dev), [c.to(dev) for c in ctxs], D.to(dev), Kfull.to(dev)
    eye = torch.eye(4, device=dev).repeat(B, 1, 1)
    # (1) identical frames + identity pose => every map is ~0 on the interior, loss ~ 0
    out = P.photometric_maps(Dd, Kd, Ad, [Ad, Ad], [eye, eye], 1.0, 1.0)
    assert out["loss"].item() < 1e-4
    # (2) auto-mask can only lower the min-reprojection loss; mean of maps >= min of maps
    p_d = [p.to(dev) for p in poses]
    l_min = P.photometric_maps(Dd, Kd, Ad, Cd, p_d, 1.0, 1.0, automask=True)["loss"].item()
    l_min_nomask = P.photometric_maps(Dd, Kd, Ad, Cd, p_d, 1.0, 1.0, automask=False)["loss"].item()
    l_mean = P.photometric_maps(Dd, Kd, Ad, Cd, p_d, 1.0, 1.0, automask=True, reduce="mean")["loss"].item()
    assert l_min <= l_min_nomask + 1e-7 and l_min <= l_mean + 1e-7
    # (3) loss equals the mean over the arg-min-selected map, and the oracle agrees on the forward value
    o = P.photometric_maps(Dd, Kd, Ad, Cd, p_d, 1.0, 1.0)
    sel_mean = o["maps"].gather(1, o["sel"].long().unsqueeze(1)).mean().item()
    assert abs(sel_mean - o["loss"].item()) < 1e-6
    loss_o, _ = _oracle_photo(A, ctxs, D, Kfull, poses, 1.0, 1.0)
    close(o["loss"], loss_o, 2e-5, 1e-6)
    # (4) backward runs at full size and is finite; zero upstream gradient gives zero
    Dg = Dd.clone().requires_grad_(True)
    pg = [p.clone().requires_grad_(True) for p in p_d]
    P.photometric_scale_loss(Dg, Kd, Ad, Cd, pg, 1.0, 1.0).backward()
    assert torch.isfinite(Dg.grad).all() and Dg.grad.abs().sum() > 0
    assert all(torch.isfinite(p.grad).all() for p in pg)


@pytest.mark.parametrize("B,h,w", [(2, 48, 160), (3, 24, 80), (1, 37, 75), (12, 192, 640)])
def test_smoothness(P, B, h, w):
    g = torch.Generator().manual_seed(h)
    img = smooth_images(g, B, h, w, 1)[0]
    D = torch.rand(B, 1, h, w, generator=g) * 40 + 0.5
    Dg = D.clone().requires_grad_(True)
    lo = OL.smoothness(Dg, img)
    (lo * 2.0).backward()
    Dd = D.clone().to(dev).requires_grad_(True)
    l = P.smoothness_loss(Dd, img.to(dev))
    (l * 2.0).backward()
    close(l, lo, 2e-5, 1e-7)
    rel = (Dd.grad.cpu().double() - Dg.grad.double()).norm() / Dg.grad.double().norm()
    assert rel < 1e-4, f"smoothness d_depth rel err {rel:.3e}"


@pytest.mark.parametrize("B,h,w,H,W", [(2, 48, 160, 48, 160), (2, 24, 80, 192, 640), (12, 96, 320, 192, 640), (1, 8, 24, 64, 192)])
def test_silog(P, B, h, w, H, W):
    g = torch.Generator().manual_seed(w)
    est = torch.rand(B, 1, h, w, generator=g) * 60 + 0.3
    gt = torch.where(torch.rand(B, 1, H, W, generator=g) < 0.3, torch.rand(B, 1, H, W, generator=g) * 79 + 1, torch.zeros(1))
    eg = est.clone().requires_grad_(True)
    lo = OL.silog(eg, G.resize_img(gt, (h, w), mode="nearest"))
    (lo * 0.25).backward()
    ed = est.clone().to(dev).requires_grad_(True)
    l = P.silog_loss(ed, gt.to(dev))
    (l * 0.25).backward()
    close(l, lo, 2e-5, 1e-6)
    close(ed.grad, eg.grad, 1e-4, 1e-9)


@pytest.mark.parametrize("B,H,W,n", [(12, 192, 640, 4), (2, 64, 192, 3), (1, 48, 160, 1)])
def test_silog_multi_scale(P, B, H, W, n):
    """silog_loss_multi = sum_k w_k silog(est_k, nearest(gt)): per-scale statistics bit-identical to the single-scale launches (same block partition and
    order), gradients equal to them, total and gradients against the oracle's per-scale loop (Supervised.py:L42-47)."""
    g = torch.Generator().manual_seed(W + n)
    gt = torch.where(torch.rand(B, 1, H, W, generator=g) < 0.3, torch.rand(B, 1, H, W, generator=g) * 79 + 1, torch.zeros(1))
    ests = [torch.rand(B, 1, H >> k, W >> k, generator=g) * 60 + 0.3 for k in range(n)]
    ws = [1.0 / n] * n
    eo = [e.clone().requires_grad_(True) for e in ests]
    lo = sum(OL.silog(e, G.resize_img(gt, e.shape[-2:], mode="nearest")) for e in eo) / n
    lo.backward()
    em = [e.clone().to(dev).requires_grad_(True) for e in ests]
    lm = P.silog_loss_multi(em, gt.to(dev), 0.85, ws)
    (lm * 3.0).backward()
    es = [e.clone().to(dev).requires_grad_(True) for e in ests]
    singles = [P.silog_loss(e, gt.to(dev), 0.85) for e in es]
    ls = singles[0] * ws[0]
    for t, w_ in zip(singles[1:], ws[1:]):
        ls = ls + t * w_
    (ls * 3.0).backward()
    assert lm.shape == () and torch.equal(lm.cpu(), ls.cpu())
    close(lm, lo, 2e-5, 1e-6)
    for a_, b_, c_ in zip(em, es, eo):
        close(a_.grad, b_.grad, 1e-6, 1e-12)
        close(a_.grad / 3.0, c_.grad, 1e-4, 1e-9)
    with pytest.raises(Exception):
        P.silog_loss_multi(em, gt.to(dev), 0.85, ws[:-1] if n > 1 else [])


def test_silog_golden(P, geo):
    for tag in ("s48", "s24"):
        D, gt = geo.t(f"{tag}.D"), geo.t(f"{tag}.gt")
        close(P.silog_loss(D.to(dev), gt.to(dev)), geo[f"{tag}.silog"], 2e-5, 1e-6)
        close(P.smoothness_loss(D.to(dev), geo.t(f"{tag}.A").to(dev)), geo[f"{tag}.smooth"], 2e-5, 1e-7)


def test_no_cpu_fallback(P):
    with pytest.raises(Exception):
        P.resize(torch.rand(1, 3, 8, 8), (4, 4))      # CPU tensor must be refused, not silently computed


@pytest.mark.parametrize("B,C,h,w", [(2, 3, 48, 160), (1, 3, 37, 53), (2, 1, 2, 2), (1, 2, 5, 130)])
def test_ssim_module_stand_alone(P, B, C, h, w):
    """SSIM()(x, y) -- the callable module of ssim_loss.py:L6-53 -- against the oracle's restatement: the map, and the gradients to BOTH images
    (reflection-border multiplicities included: the 2x2 and 5-row cases are all border)."""
    from simpledepthestimation_amd.modeling.losses.ssim_loss import SSIM
    g = torch.Generator().manual_seed(B * 100 + h)
    x = torch.rand(B, C, h, w, generator=g)
    y = (x + 0.15 * torch.randn(B, C, h, w, generator=g)).clamp(0, 1)
    xr, yr = x.clone().requires_grad_(True), y.clone().requires_grad_(True)
    ref = OL.ssim_distance(xr, yr)
    go = torch.rand(ref.shape, generator=g)
    ref.backward(go)
    xd, yd = x.to(dev).requires_grad_(True), y.to(dev).requires_grad_(True)
    out = SSIM()(xd, yd)
    assert out.shape == ref.shape
    # (E[x^2] - mu^2 in fp32 with random, low-variance windows: the two fp32 evaluations differ by cancellation noise, not by formula)
    assert torch.allclose(out.cpu(), ref.detach(), rtol=2e-4, atol=2e-5), (out.cpu() - ref.detach()).abs().max()
    out.backward(go.to(dev))
    for name, got, want in (("dx", xd.grad, xr.grad), ("dy", yd.grad, yr.grad)):
        err = (got.cpu() - want).abs().max().item()
        assert err < 2e-3 * want.abs().max().item() + 1e-6, f"{name}: max abs error {err:.3e} (scale {want.abs().max().item():.3e})"
        rel = ((got.cpu().double() - want.double()).norm() / want.double().norm()).item()
        assert rel < 1e-3, f"{name}: relative L2 error {rel:.3e}"
    # only one input needs a gradient
    xd2 = x.to(dev).requires_grad_(True)
    SSIM()(xd2, y.to(dev)).backward(go.to(dev))
    assert torch.equal(xd2.grad, xd.grad)


def test_smoothness_reversed_is_the_same_loss(P):
    """smoothness_loss(reversed=True) flips every finite difference, which only enter through abs(): same value, same gradient."""
    from simpledepthestimation_amd.modeling.losses.smoothness_loss import smoothness_loss
    g = torch.Generator().manual_seed(3)
    depth = torch.rand(2, 1, 24, 80, generator=g) * 20 + 1
    image = torch.rand(2, 3, 24, 80, generator=g)
    d0 = depth.to(dev).requires_grad_(True); d1 = depth.to(dev).requires_grad_(True)
    l0 = smoothness_loss(d0, image.to(dev)); l1 = smoothness_loss(d1, image.to(dev), reversed=True)
    l0.backward(); l1.backward()
    ref0 = OL.smoothness(depth, image)
    assert torch.equal(l0, l1) and torch.equal(d0.grad, d1.grad)
    assert abs(l0.item() - ref0.item()) < 2e-5 * abs(ref0.item()) + 1e-8


def test_pose_utils_euler2mat_and_invert_pose(P):
    """geometry/pose_utils.py on the device: euler2mat = the rotation block of sde_pose_vec2mat, invert_pose; fixtures from the reference's functions."""
    import os
    from conftest import GOLDEN, _Golden
    from simpledepthestimation_amd.geometry import pose_utils as PU
    gd = _Golden(os.path.join(GOLDEN, "data.npz"))
    R = PU.euler2mat(torch.from_numpy(gd["pu.angles"]).cuda())
    assert tuple(R.shape) == (6, 3, 3) and np.allclose(R.cpu().numpy(), gd["pu.euler2mat"], rtol=0, atol=2e-6)
    T = torch.from_numpy(gd["pu.T"].astype(np.float32)).cuda()
    Ti = PU.invert_pose(T)
    assert np.allclose(Ti.cpu().numpy(), gd["pu.invert_pose"], rtol=0, atol=2e-6)
    assert torch.allclose(Ti @ T, torch.eye(4, device="cuda").expand(6, 4, 4), atol=1e-5)
