"""PackNet's 3-D convolution at the ten shapes of the PackNet-1A step (bs 12, 192x640): forward / data gradient / weight gradient times through the C ABI.
A/B builds: SDE_HIP_LIB=/path/to/other.so python scripts/microbench_conv3d.py"""
import sys

import torch

sys.path.insert(0, ".")
from simpledepthestimation_amd.hip import lib as L, nn as HN  # noqa: E402

SHAPES = [(12, 96, 320, 256), (12, 48, 160, 256), (12, 24, 80, 512), (12, 12, 40, 1024), (12, 6, 20, 2048),
          (12, 6, 20, 256), (12, 12, 40, 128), (12, 24, 80, 64), (12, 48, 160, 32), (12, 96, 320, 32)]
dev = "cuda"
print("lib:", L.LIB_PATH)
tot = [0.0, 0.0, 0.0]
for B, H, W, D in SHAPES:
    x = torch.randn(B, H, W, D, device=dev).bfloat16()
    w = torch.randn(8, 1, 3, 3, 3, device=dev) * 0.2
    b = torch.zeros(8, device=dev)
    y = torch.empty(B, H, W, 8 * D, device=dev, dtype=torch.bfloat16)
    dy = torch.randn(B, H, W, 8 * D, device=dev).bfloat16()
    dx = torch.empty_like(x)
    lib = L.lib()
    dc = HN.dtype_code(x.dtype)
    part = torch.empty(lib.sde_conv3d_wgrad_num_blocks(B, H, W, D, dc), 224, device=dev)
    dw, db = torch.empty_like(w), torch.empty_like(b)
    calls = [lambda: lib.sde_conv3d_fwd(L.ptr(x), L.ptr(w), L.ptr(b), B, H, W, D, dc, L.ptr(y), L.stream()),
             lambda: lib.sde_conv3d_dgrad(L.ptr(dy), L.ptr(w), B, H, W, D, dc, L.ptr(dx), L.stream()),
             lambda: lib.sde_conv3d_wgrad(L.ptr(x), L.ptr(dy), B, H, W, D, dc, L.ptr(part), L.ptr(dw), L.ptr(db), 0, L.stream())]
    ts = []
    for c in calls:
        for _ in range(2):
            c()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(5):
            c()
        e1.record()
        torch.cuda.synchronize()
        ts.append(e0.elapsed_time(e1) / 5 * 1e3)
    for i in range(3):
        tot[i] += ts[i]
    gb = x.numel() * 2 * 9 / 1e9
    print(f"B{B} {H:3d}x{W:3d} D{D:5d}: fwd {ts[0]:8.1f} us  dgrad {ts[1]:8.1f} us  wgrad {ts[2]:8.1f} us   (in+out {gb:.2f} GB)")
print(f"sum: fwd {tot[0] / 1e3:.2f} ms  dgrad {tot[1] / 1e3:.2f} ms  wgrad {tot[2] / 1e3:.2f} ms")
