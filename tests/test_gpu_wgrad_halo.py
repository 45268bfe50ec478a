"""GPU parity of the LDS-halo weight-gradient kernel (csrc/wgrad_halo.hip) through the C ABI.

The decoder's high-resolution 3x3 layers (<= 96 input, <= 32 output channels; depth_decoder.py:L21-53, L95-110) are compared with
  * plain torch-CPU fp32 convolution backward on the 16-bit-rounded operands, and
  * the generic weight-gradient kernel on the same device buffers (SDE_OPT_WGRAD_HALO = 0): both accumulate in fp32, so they differ
    by summation order only.
Covers every (input blocks, output blocks) instantiation, reflect and zero padding, the up-sampled and the up-sampled + concatenated
source, ragged tiles (sizes that are not multiples of the 8 x 32 tile), Cout = 1 (the disparity heads) and fp16.
"""
import ctypes
import math

import pytest
import torch
import torch.nn.functional as F

pytestmark = pytest.mark.gpu
dev = "cuda"

CASES = [
    # name, B, H, W (input of the conv = output size), C0, C1, Cout, reflect, upcat, dtype
    ("up16_16", 2, 96, 96, 16, 0, 16, True, True, torch.bfloat16),
    ("up32_64_32", 2, 96, 128, 32, 64, 32, True, True, torch.bfloat16),
    ("plain32_16_ragged", 3, 77, 75, 32, 0, 16, True, False, torch.bfloat16),
    ("plain64_32", 2, 96, 96, 64, 0, 32, True, False, torch.bfloat16),
    ("disp16_1", 2, 96, 96, 16, 0, 1, True, False, torch.bfloat16),
    ("disp32_1_ragged", 2, 100, 90, 32, 0, 1, True, False, torch.bfloat16),
    ("disp64_1", 2, 96, 96, 64, 0, 1, True, False, torch.bfloat16),
    ("zero16_32", 2, 90, 100, 16, 0, 32, False, False, torch.bfloat16),
    ("zero32_32", 2, 96, 96, 32, 0, 32, False, False, torch.bfloat16),
    ("up32_64_16", 2, 96, 96, 32, 64, 16, True, True, torch.bfloat16),
    ("up16_16_fp16", 2, 96, 96, 16, 0, 16, True, True, torch.float16),
    ("plain64_32_fp16", 2, 96, 96, 64, 0, 32, True, False, torch.float16),
]


@pytest.mark.parametrize("case", CASES, ids=[c[0] for c in CASES])
def test_wgrad_halo(case):
    from simpledepthestimation_amd.hip import lib as L
    from simpledepthestimation_amd.hip import nn as NN
    name, B, H, W, C0, C1, Cout, reflect, upcat, dt = case
    g = torch.Generator().manual_seed(len(name) * 13 + B)
    h0, w0 = (H // 2, W // 2) if upcat else (H, W)
    x0 = torch.randn(B, C0, h0, w0, generator=g).to(dt).float()
    x1 = torch.randn(B, C1, H, W, generator=g).to(dt).float() if C1 else None
    Cin = C0 + C1
    wt = torch.randn(Cout, Cin, 3, 3, generator=g) / math.sqrt(Cin * 9)
    xin = x0
    if upcat:
        up = F.interpolate(x0, scale_factor=2, mode="nearest")
        xin = torch.cat([up, x1], 1) if C1 else up
    if reflect:
        xin = F.pad(xin, (1, 1, 1, 1), mode="reflect")
    wr = wt.clone().requires_grad_(True)
    yr = F.conv2d(xin, wr, None, 1, 0 if reflect else 1)
    gy = torch.randn(yr.shape, generator=g).to(dt).float()
    yr.backward(gy)

    def nhwc(t):
        return t.permute(0, 2, 3, 1).contiguous().to(dt).to(dev)

    res = {}
    for on in (1, 0):
        old = NN.set_option(NN.OPT_WGRAD_HALO, on)
        try:
            xd = nhwc(x0)
            x1d = nhwc(x1) if C1 else None
            wd = wt.clone().to(dev).requires_grad_(True)
            y = NN.conv2d(xd, wd, None, stride=1, pad=1, reflect=reflect, act=0, skip=x1d, upsample=upcat)
            d = NN._desc(xd, x1d, NN.SRC_UPCAT if upcat else NN.SRC_PLAIN, 3, 3, 1, 1, reflect, H, W, H, W)
            splits = L.lib().sde_conv_wgrad_splits(ctypes.byref(d), Cout)
            gyd = nhwc(gy)
            y.backward(gyd if y.shape[3] == Cout else F.pad(gyd, (0, y.shape[3] - Cout)))
            torch.cuda.synchronize()
            res[on] = (wd.grad.detach().cpu().double(), splits)
        finally:
            NN.set_option(NN.OPT_WGRAD_HALO, old)
    (dw1, s1), (dw0, s0) = res[1], res[0]
    tiles = B * ((H + 7) // 8) * ((W + 31) // 32)
    assert s1 == min(tiles, s1) and s1 in (min(tiles, 256), min(tiles, 512)), (s1, tiles)      # one slab per persistent workgroup
    ref = wr.grad.double()
    e1 = ((dw1 - ref).norm() / ref.norm()).item()
    e0 = ((dw0 - ref).norm() / ref.norm()).item()
    e10 = ((dw1 - dw0).norm() / dw0.norm()).item()
    print(f"{name}: splits halo {s1} generic {s0}; rel L2 vs fp32 CPU: halo {e1:.2e} generic {e0:.2e}; halo vs generic {e10:.2e}")
    assert e1 < 1e-5 and e10 < 1e-5, (e1, e0, e10)
    assert torch.isfinite(dw1).all()
