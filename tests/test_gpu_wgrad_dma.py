"""GPU parity of the LDS-DMA weight-gradient kernel (csrc/wgrad_dma.hip) through the C ABI.

Layers with 64-channel-multiple inputs and outputs and zero padding (the ResNet encoders' 1x1 and 3x3 convolutions, stride 1 and 2;
resnet_encoder.py:L88-99) are compared with
  * plain torch-CPU fp32 convolution backward on the 16-bit-rounded operands, and
  * the register-staged weight-gradient kernel on the same device buffers (SDE_OPT_WGRAD_DMA = 0): fp32 accumulation both, summation order only.
Covers 1x1 / 3x3 / 5x5, strides 1 and 2, reflection padding, the up-sampled (+ concatenated) source of the decoder, ragged pixel ranges (M not a multiple of 64, odd sizes), an odd number of 64-column K blocks (the second
half of the last tile is empty), several output-channel tiles, pixel splits (slab stacks) and the single-split direct case, and fp16.
"""
import ctypes
import math

import pytest
import torch
import torch.nn.functional as F

pytestmark = pytest.mark.gpu
dev = "cuda"

CASES = [
    # name, B, H, W (input), Cin, Cout, k, stride, pad, dtype
    ("1x1_64_256", 4, 48, 80, 64, 256, 1, 1, 0, torch.bfloat16),
    ("1x1_256_64", 4, 48, 80, 256, 64, 1, 1, 0, torch.bfloat16),
    ("1x1_1024_256_small", 2, 12, 40, 1024, 256, 1, 1, 0, torch.bfloat16),
    ("3x3_64_64", 4, 48, 80, 64, 64, 3, 1, 1, torch.bfloat16),
    ("3x3_128_128_ragged", 3, 23, 41, 128, 128, 3, 1, 1, torch.bfloat16),
    ("3x3_s2_128_128", 2, 48, 80, 128, 128, 3, 2, 1, torch.bfloat16),
    ("3x3_s2_64_128_odd", 2, 13, 21, 64, 128, 3, 2, 1, torch.bfloat16),
    ("1x1_s2_256_512", 2, 48, 80, 256, 512, 1, 2, 0, torch.bfloat16),
    ("5x5_64_64", 1, 20, 36, 64, 64, 5, 1, 2, torch.bfloat16),
    ("3x3_512_512_tinyM", 2, 6, 20, 512, 512, 3, 1, 1, torch.bfloat16),
    ("3x3_64_64_fp16", 2, 48, 80, 64, 64, 3, 1, 1, torch.float16),
    ("1x1_128_512_fp16", 2, 24, 80, 128, 512, 1, 1, 0, torch.float16),
    # decoder: reflection padding, and the nearest x2 up-sampled + concatenated source (Cin = C0 up-sampled + C1 skip channels)
    ("refl_128_64", 2, 24, 80, 128, 64, 3, 1, 1, torch.bfloat16, True, 0),
    ("refl_256_128_ragged", 3, 13, 41, 256, 128, 3, 1, 1, torch.bfloat16, True, 0),
    ("upcat_64_256_64", 2, 48, 80, 64, 64, 3, 1, 1, torch.bfloat16, True, 256),
    ("upcat_128_0_128", 2, 24, 40, 128, 128, 3, 1, 1, torch.bfloat16, True, -1),
    ("upcat_128_512_128_fp16", 1, 24, 80, 128, 128, 3, 1, 1, torch.float16, True, 512),
]


@pytest.mark.parametrize("case", CASES, ids=[c[0] for c in CASES])
def test_wgrad_dma(case):
    from simpledepthestimation_amd.hip import nn as NN
    name, B, H, W, C0, Cout, k, stride, pad, dt = case[:10]
    reflect, C1 = (case[10], case[11]) if len(case) > 10 else (False, 0)
    upcat = C1 != 0                            # C1 = -1: up-sampled source without a skip tensor
    C1 = max(C1, 0)
    Cin = C0 + C1
    g = torch.Generator().manual_seed(len(name) * 11 + B)
    x = torch.randn(B, C0, (H // 2, W // 2)[0] if upcat else H, (H // 2, W // 2)[1] if upcat else W, generator=g).to(dt).float()
    x1 = torch.randn(B, C1, H, W, generator=g).to(dt).float() if C1 else None
    wt = torch.randn(Cout, Cin, k, k, generator=g) / math.sqrt(Cin * k * k)
    wr = wt.clone().requires_grad_(True)
    xin = x
    if upcat:
        xin = F.interpolate(x, scale_factor=2, mode="nearest")
        if C1:
            xin = torch.cat([xin, x1], 1)
    if reflect:
        xin = F.pad(xin, (1, 1, 1, 1), mode="reflect")
    yr = F.conv2d(xin, wr, None, stride, 0 if reflect else pad)
    gy = torch.randn(yr.shape, generator=g).to(dt).float()
    yr.backward(gy)

    def nhwc(t):
        return t.permute(0, 2, 3, 1).contiguous().to(dt).to(dev)

    res = {}
    for on in (1, 0):
        old = NN.set_option(NN.OPT_WGRAD_DMA, on)
        try:
            xd = nhwc(x)
            x1d = nhwc(x1) if C1 else None
            wd = wt.clone().to(dev).requires_grad_(True)
            y = NN.conv2d(xd, wd, None, stride=stride, pad=pad, reflect=reflect, skip=x1d, upsample=upcat)
            y.backward(nhwc(gy))
            torch.cuda.synchronize()
            res[on] = wd.grad.detach().cpu().double()
        finally:
            NN.set_option(NN.OPT_WGRAD_DMA, old)
    ref = wr.grad.double()
    e1 = ((res[1] - ref).norm() / ref.norm()).item()
    e0 = ((res[0] - ref).norm() / ref.norm()).item()
    e10 = ((res[1] - res[0]).norm() / res[0].norm()).item()
    print(f"{name}: rel L2 vs fp32 CPU: LDS-DMA {e1:.2e} register-staged {e0:.2e}; one vs the other {e10:.2e}")
    assert torch.isfinite(res[1]).all()
    assert e1 < 1e-5 and e10 < 1e-5, (e1, e0, e10)
