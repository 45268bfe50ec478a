"""Per-shape timing of the convolution engine (forward, dgrad, wgrad) -- used for A/B comparisons of kernel variants.
   SDE_HIP_LIB=<path to .so> python scripts/microbench_conv.py"""
import os, sys, math
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from simpledepthestimation_amd.hip import nn as HN, lib as L

SHAPES = [  # name, B, H, W, Cin, Cout, k, stride, pad, reflect, upcat(C1)
    ("l1_3x3_64_64", 12, 48, 160, 64, 64, 3, 1, 1, False, 0),
    ("l1_1x1_64_256", 12, 48, 160, 64, 256, 1, 1, 0, False, 0),
    ("l1_1x1_256_64", 12, 48, 160, 256, 64, 1, 1, 0, False, 0),
    ("l2_3x3_128_128", 12, 24, 80, 128, 128, 3, 1, 1, False, 0),
    ("l3_3x3_256_256", 12, 12, 40, 256, 256, 3, 1, 1, False, 0),
    ("l3_1x1_1024_256", 12, 12, 40, 1024, 256, 1, 1, 0, False, 0),
    ("l4_3x3_512_512", 12, 6, 20, 512, 512, 3, 1, 1, False, 0),
    ("dec_up31_640_128", 12, 12, 40, 128, 128, 3, 1, 1, True, 512),
    ("dec_up21_320_64", 12, 24, 80, 64, 64, 3, 1, 1, True, 256),
    ("dec_up11_96_32", 12, 48, 160, 32, 32, 3, 1, 1, True, 64),
    ("dec_up01_16_16", 12, 96, 320, 16, 16, 3, 1, 1, True, 0),
    ("stem_7x7", 12, 192, 640, 3, 64, 7, 2, 3, False, 0),
    ("l2_1x1_128_512", 12, 24, 80, 128, 512, 1, 1, 0, False, 0),
    ("l2_1x1_512_128", 12, 24, 80, 512, 128, 1, 1, 0, False, 0),
    ("l3_1x1_256_1024", 12, 12, 40, 256, 1024, 1, 1, 0, False, 0),
    ("l4_1x1_512_2048", 12, 6, 20, 512, 2048, 1, 1, 0, False, 0),
    ("l4_1x1_2048_512", 12, 6, 20, 2048, 512, 1, 1, 0, False, 0),
    ("l2_3x3s2_128_128", 12, 48, 160, 128, 128, 3, 2, 1, False, 0),
    ("dec_up40_2048_256", 12, 6, 20, 2048, 256, 3, 1, 1, True, 0),
    ("dec_up41_1280_256", 12, 6, 20, 256, 256, 3, 1, 1, True, 1024),
    ("dec_up30_256_128", 12, 12, 40, 256, 128, 3, 1, 1, True, 0),
    ("dec_up20_128_64", 12, 24, 80, 128, 64, 3, 1, 1, True, 0),
    ("dec_up10_64_32", 12, 48, 160, 64, 32, 3, 1, 1, True, 0),
]
for key, env in ((HN.OPT_PGEMM, "PG"), (HN.OPT_PGEMM_DEPTH, "PGD"), (HN.OPT_PGEMM_3X3, "PG3"), (HN.OPT_WGRAD_DMA, "WD"), (HN.OPT_WGRAD_BLOCKS, "WB")):
    if os.environ.get(env) is not None:
        HN.set_option(key, int(os.environ[env]))
print("options:", {e: os.environ.get(e) for e in ("PG", "PGD", "PG3", "WD", "WB")})
dt = torch.bfloat16
dev = "cuda"
print("lib:", L.LIB_PATH)
print(f"{'shape':20s} {'fwd us':>8s} {'TF/s':>7s} {'dgrad us':>9s} {'TF/s':>7s} {'wgrad us':>9s} {'TF/s':>7s}")
FILTER = os.environ.get("MB_FILTER")          # substring of the shape name
STATS = bool(int(os.environ.get("MB_STATS", "0")))   # forward with the BatchNorm partial sums, as the encoder layers run
for name, B, H, W, Cin, Cout, k, s, p, refl, C1 in SHAPES:
    if FILTER and FILTER not in name:
        continue
    g = torch.Generator().manual_seed(0)
    Cp = (Cin + 7) // 8 * 8
    x = torch.randn(B, H, W, Cp, generator=g).to(dt).to(dev)
    skip = torch.randn(B, 2 * H, 2 * W, C1, generator=g).to(dt).to(dev) if C1 else None
    up = refl and (C1 > 0 or name.startswith("dec_up0"))
    w = (torch.randn(Cout, Cin + C1, k, k, generator=g) / math.sqrt((Cin + C1) * k * k)).to(dev).requires_grad_(True)
    xg = x.clone().requires_grad_(Cin >= 8)
    sg = skip.clone().requires_grad_(True) if skip is not None else None
    def fwd():
        r = HN.conv2d(xg, w, None, stride=s, pad=p, reflect=refl, skip=sg, upsample=up, bn_stats=STATS)
        return r[0] if isinstance(r, tuple) else r
    y = fwd()
    gy = torch.randn_like(y)
    res = {}
    for _ in range(3):
        L.PROFILE, L.PROFILE_REPEAT = [], 6        # six launches per event pair (lib.timed): the pair itself costs ~10 us
        for _ in range(5):
            y = fwd()
            y.backward(gy)
            w.grad = None
        torch.cuda.synchronize()
        recs, L.PROFILE = L.PROFILE, None
        for kind, flops, variant, e0, e1, meta, rep in recs:
            res.setdefault(kind, []).append((e0.elapsed_time(e1) * 1e3 / rep, flops))
    def stat(kind):
        if kind not in res: return (0.0, 0.0)
        us = sorted(r[0] for r in res[kind])[len(res[kind]) // 2]
        return us, res[kind][0][1] / us / 1e6
    f, d, wg = stat("igemm_fwd"), stat("igemm_dgrad"), stat("wgrad")
    print(f"{name:20s} {f[0]:8.1f} {f[1]:7.1f} {d[0]:9.1f} {d[1]:7.1f} {wg[0]:9.1f} {wg[1]:7.1f}")
