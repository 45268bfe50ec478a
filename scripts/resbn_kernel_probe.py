"""Kernel-level check of sde_conv_dgrad_bnbwd_res against torch on the GPU: where do gm and the partial sums differ?"""
import ctypes, sys
import torch
sys.path.insert(0, ".")
from simpledepthestimation_amd.hip import lib as L, nn as HN

dev = "cuda"
lib = L.lib()
for (B, H, W, N, K) in [(2, 24, 40, 256, 64), (2, 12, 20, 512, 128), (12, 48, 160, 256, 64), (1, 7, 9, 256, 64)]:
    g = torch.Generator().manual_seed(N + K)
    dt = torch.bfloat16
    M = B * H * W
    dz = (torch.randn(B, H, W, K, generator=g) * 0.3).to(dt).to(dev)
    w = (torch.randn(K, N, 1, 1, generator=g) / 8).to(dev)                       # conv1: N -> K channels; its data gradient is [M, N]
    y = torch.randn(B, H, W, N, generator=g).to(dt).to(dev)
    go = (torch.randn(B, H, W, N, generator=g) * 0.2).to(dt).to(dev)
    out = torch.relu(torch.randn(B, H, W, N, generator=g)).to(dt).to(dev)
    bnp = torch.stack([torch.randn(N, generator=g) * 0.1, torch.rand(N, generator=g) + 0.5, torch.rand(N, generator=g) + 0.5, torch.randn(N, generator=g) * 0.1]).to(dev).contiguous()
    wd = HN.pack_weight(w, dt, N, K, for_dgrad=True)
    dd = HN._desc(dz, None, HN.SRC_PLAIN, 1, 1, 1, 0, False, H, W, H, W)
    rows = lib.sde_conv_dgrad_bnbwd_res_rows(ctypes.byref(dd), N, N)
    part = torch.full((rows + HN.REDUCE_ROWS, N, 2), float("nan"), device=dev)
    gm = torch.full((B, H, W, N), float("nan"), device=dev, dtype=dt)
    L.check(lib.sde_conv_dgrad_bnbwd_res(ctypes.byref(dd), L.ptr(wd), L.ptr(gm), N, N, L.ptr(y), L.ptr(bnp), L.ptr(part), L.ptr(go), L.ptr(out), L.stream()), "res")
    torch.cuda.synchronize()
    g_ref = (dz.float().reshape(M, K) @ w.reshape(K, N).to(dt).float()).to(dt).float()          # the data gradient, rounded to storage
    s = (g_ref + go.float().reshape(M, N)).to(dt).float()
    gm_ref = torch.where(out.float().reshape(M, N) > 0, s, torch.zeros_like(s))
    xhat = (y.float().reshape(M, N) - bnp[0]) * bnp[1]
    err = (gm.float().reshape(M, N) - gm_ref).abs()
    bad = err > 2e-2 * gm_ref.abs().max()
    s1, s2 = part[:rows, :, 0].sum(0), part[:rows, :, 1].sum(0)
    print(f"B{B} {H}x{W} N{N} K{K}: rows {rows}; gm bad {int(bad.sum())}/{bad.numel()} (nan {int(torch.isnan(gm.float()).sum())}); "
          f"bad rows {sorted(set((bad.nonzero()[:, 0] % 64).tolist()))[:16]} bad cols mod 64 {sorted(set((bad.nonzero()[:, 1] % 64).tolist()))[:16]}; "
          f"sum gm rel {float((s1 - gm_ref.sum(0)).norm() / gm_ref.sum(0).norm()):.2e}  sum gm*xhat rel {float((s2 - (gm_ref * xhat).sum(0)).norm() / (gm_ref * xhat).sum(0).norm()):.2e}  part nan {int(torch.isnan(part[:rows]).sum())}")
