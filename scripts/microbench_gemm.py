"""A/B timing of the forward / data-gradient GEMM kernels on the layer shapes of the benchmark workloads, through the C ABI.

Every configuration is timed as REP back-to-back launches between two events (event-pair overhead, ~8 us, would otherwise swamp 10-30 us
kernels), best of ROUNDS interleaved rounds in ONE process (cdna_hip_programming.md rule 24).  Configurations = dispatcher options
(sde_conv_set_option): "old" = register-staged kernels, "pg4" / "pg3" = persistent LDS-DMA GEMM with a 4 / 3 stage ring, "+3" = also on the
layers the LDS-halo kernel takes.   python scripts/microbench_gemm.py [r50|r18] [configs...]"""
import ctypes, math, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from simpledepthestimation_amd.hip import nn as HN, lib as L

REP, ROUNDS = 20, 5
dev = "cuda"
dt = torch.bfloat16


def layer_table(enc):
    """(name, H, W, C0, C1, Cout, k, stride, pad, reflect) of the distinct conv shapes at 192x640 (B = 12): encoder + decoder."""
    T = []
    if enc == 50:
        T += [("l1.c1 64>64", 48, 160, 64, 0, 64, 1, 1, 0, 0), ("l1.c1 256>64", 48, 160, 256, 0, 64, 1, 1, 0, 0), ("l1.c2 3x3 64", 48, 160, 64, 0, 64, 3, 1, 1, 0),
              ("l1.c3 64>256", 48, 160, 64, 0, 256, 1, 1, 0, 0),
              ("l2.c1 256>128", 48, 160, 256, 0, 128, 1, 1, 0, 0), ("l2.c2 3x3s2 128", 48, 160, 128, 0, 128, 3, 2, 1, 0), ("l2.ds 256>512s2", 48, 160, 256, 0, 512, 1, 2, 0, 0),
              ("l2.c1 512>128", 24, 80, 512, 0, 128, 1, 1, 0, 0), ("l2.c2 3x3 128", 24, 80, 128, 0, 128, 3, 1, 1, 0), ("l2.c3 128>512", 24, 80, 128, 0, 512, 1, 1, 0, 0),
              ("l3.c1 512>256", 24, 80, 512, 0, 256, 1, 1, 0, 0), ("l3.c2 3x3s2 256", 24, 80, 256, 0, 256, 3, 2, 1, 0), ("l3.ds 512>1024s2", 24, 80, 512, 0, 1024, 1, 2, 0, 0),
              ("l3.c1 1024>256", 12, 40, 1024, 0, 256, 1, 1, 0, 0), ("l3.c2 3x3 256", 12, 40, 256, 0, 256, 3, 1, 1, 0), ("l3.c3 256>1024", 12, 40, 256, 0, 1024, 1, 1, 0, 0),
              ("l4.c1 1024>512", 12, 40, 1024, 0, 512, 1, 1, 0, 0), ("l4.c2 3x3s2 512", 12, 40, 512, 0, 512, 3, 2, 1, 0), ("l4.ds 1024>2048s2", 12, 40, 1024, 0, 2048, 1, 2, 0, 0),
              ("l4.c1 2048>512", 6, 20, 2048, 0, 512, 1, 1, 0, 0), ("l4.c2 3x3 512", 6, 20, 512, 0, 512, 3, 1, 1, 0), ("l4.c3 512>2048", 6, 20, 512, 0, 2048, 1, 1, 0, 0)]
        enc_ch = [64, 256, 512, 1024, 2048]
    else:
        T += [("l1 3x3 64", 48, 160, 64, 0, 64, 3, 1, 1, 0), ("l2 3x3s2 64>128", 48, 160, 64, 0, 128, 3, 2, 1, 0), ("l2 3x3 128", 24, 80, 128, 0, 128, 3, 1, 1, 0),
              ("l3 3x3s2 128>256", 24, 80, 128, 0, 256, 3, 2, 1, 0), ("l3 3x3 256", 12, 40, 256, 0, 256, 3, 1, 1, 0),
              ("l4 3x3s2 256>512", 12, 40, 256, 0, 512, 3, 2, 1, 0), ("l4 3x3 512", 6, 20, 512, 0, 512, 3, 1, 1, 0)]
        enc_ch = [64, 64, 128, 256, 512]
    dec = [16, 32, 64, 128, 256]
    for i in range(4, -1, -1):
        h, w = 6 * 2 ** (4 - i), 20 * 2 ** (4 - i)
        cin = enc_ch[4] if i == 4 else dec[i + 1]
        T.append((f"up{i}0 {cin}>{dec[i]}", h, w, cin, 0, dec[i], 3, 1, 1, 1))
        skip = enc_ch[i - 1] if i > 0 else 0
        T.append((f"up{i}1 {dec[i]}+{skip}>{dec[i]}", h, w, dec[i], skip, dec[i], 3, 1, 1, 2))      # reflect = 2: upsample + concat source
    return T


CONFIGS = {"old": (0, 4, 0, 0), "pg4": (1, 4, 0, 0), "pg3": (1, 3, 0, 0), "pg4+3": (1, 4, 1, 0), "pg3+3": (1, 3, 1, 0),
           "pg3s": (1, 3, 1, 64064), "pg4s": (1, 4, 1, 64064), "pg3m": (1, 3, 1, 128064), "pg3l": (1, 3, 1, 128128)}      # forced tiles (small / mid / large)


SPLITK = int(os.environ.get("MBG_SPLITK", "1"))      # SDE_OPT_SPLITK value for every configuration (>= 2: stage threshold of the 2-way split of 256-511-tile layers)


def apply(cfg):
    HN.set_option(HN.OPT_SPLITK, SPLITK)
    on, depth, t3, tile = CONFIGS[cfg]
    HN.set_option(HN.OPT_PGEMM, on); HN.set_option(HN.OPT_PGEMM_DEPTH, depth); HN.set_option(HN.OPT_PGEMM_3X3, t3); HN.set_option(HN.OPT_PGEMM_TILE, tile)


def main():
    enc = 18 if "r18" in sys.argv[1:] else 50
    cfgs = [a for a in sys.argv[1:] if a in CONFIGS] or ["old", "pg4", "pg3", "pg3+3"]
    B = 12
    lib = L.lib()
    g = torch.Generator().manual_seed(0)
    print(f"ResNet-{enc}, B={B}, 192x640, bf16; us per launch (best of {ROUNDS} x {REP} back-to-back), TFLOP/s of the best config")
    print(f"{'layer':24s} {'kind':6s} " + " ".join(f"{c:>8s}" for c in cfgs) + "   variant(old -> last)   best TF/s")
    tot = {c: 0.0 for c in cfgs}
    for name, H, W, C0, C1, Cout, k, stride, pad, refl in layer_table(enc):
        upcat = refl == 2
        if C0 % 8:
            continue
        x0 = torch.randn(B, H, W, C0, generator=g).to(dt).to(dev)
        x1 = torch.randn(B, 2 * H, 2 * W, C1, generator=g).to(dt).to(dev) if C1 else None
        Cin = C0 + C1
        IH, IW = (2 * H, 2 * W) if upcat else (H, W)
        OH, OW = (IH + 2 * pad - k) // stride + 1, (IW + 2 * pad - k) // stride + 1
        w = (torch.randn(Cout, Cin, k, k, generator=g) / math.sqrt(Cin * k * k)).to(dev)
        ldy = HN.pad_to(Cout, 8)
        jobs = []
        d = HN._desc(x0, x1, HN.SRC_UPCAT if upcat else HN.SRC_PLAIN, k, k, stride, pad, bool(refl), IH, IW, OH, OW)
        jobs.append(("fwd", d, HN.pack_weight(w, dt, Cin, ldy), Cout, ldy, (x0, x1)))
        dz = torch.randn(B, OH, OW, ldy, generator=g).to(dt).to(dev)
        wd = HN.pack_weight(w, dt, Cin, ldy, for_dgrad=True)
        if refl:
            dd = HN._desc(dz, None, HN.SRC_PLAIN, k, k, 1, k - 1, False, OH, OW, IH + 2, IW + 2)
        elif stride == 1:
            dd = HN._desc(dz, None, HN.SRC_PLAIN, k, k, 1, k - 1 - pad, False, OH, OW, IH, IW)
        else:
            dd = HN._desc(dz, None, HN.SRC_ZEROINS, k, k, 1, k - 1 - pad, False, 2 * OH - 1, 2 * OW - 1, IH, IW)
        jobs.append(("dgrad", dd, wd, Cin, Cin, (dz,)))
        flops = 2.0 * B * OH * OW * Cout * k * k * Cin
        for kind, dsc, wp, co, ld, keep in jobs:
            y = torch.empty(dsc.Bn, dsc.OH, dsc.OW, ld, device=dev, dtype=dt)
            best, variants = {}, {}
            for c in cfgs:
                apply(c)
                variants[c] = lib.sde_conv_fwd_variant(ctypes.byref(dsc), ld)
            wsb = 0
            for c in cfgs:
                apply(c)
                wsb = max(wsb, lib.sde_conv_fwd_ws_bytes(ctypes.byref(dsc), ld))
            ws = torch.empty(max(wsb, 16) // 4, device=dev)
            for rnd in range(ROUNDS):
                for c in cfgs:
                    apply(c)
                    wsb_c = lib.sde_conv_fwd_ws_bytes(ctypes.byref(dsc), ld)
                    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                    e0.record()
                    for _ in range(REP):
                        L.check(lib.sde_conv_fwd_ws(ctypes.byref(dsc), L.ptr(wp), None, 0, L.ptr(y), co, ld, None, L.ptr(ws) if wsb_c else None, wsb_c, L.stream()), "conv")
                    e1.record(); e1.synchronize()
                    us = e0.elapsed_time(e1) * 1e3 / REP
                    best[c] = min(best.get(c, 1e9), us)
            for c in cfgs:
                tot[c] += best[c]
            bc = min(cfgs, key=lambda c: best[c])
            print(f"{name:24s} {kind:6s} " + " ".join(f"{best[c]:8.1f}" for c in cfgs) + f"   {variants[cfgs[0]]} -> {variants[cfgs[-1]]}   {flops / best[bc] / 1e6:7.1f} ({bc})")
    print(f"{'sum':24s} {'':6s} " + " ".join(f"{tot[c]:8.1f}" for c in cfgs))
    apply("pg4")


if __name__ == "__main__":
    main()
