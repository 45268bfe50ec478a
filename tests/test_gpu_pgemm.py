"""GPU parity of the persistent LDS-DMA GEMM (csrc/pgemm.hip) through the C ABI.

Every case is a convolution the dispatcher routes to that kernel (bf16, 64-channel-multiple inputs).  It is compared
  * with plain torch-CPU fp32 convolution on the bf16-rounded operands (tolerance = bf16 output rounding), and
  * with the register-staged kernels of conv.hip on the same device buffers (SDE_OPT_PGEMM = 0): both accumulate in fp32, so they
    may differ only by summation order -- at most one bf16 ulp per element, and the BatchNorm partial sums to 1e-3 relative.
Covers the four source kinds, all three tiles (64x64, 128x64, 128x128), both ring depths, ragged M / N, bias + ELU in the epilogue,
BatchNorm statistics, split-K and the data gradient (reference: resnet_encoder.py:L88-99, depth_decoder.py:L21-53,L95-110).
"""
import ctypes
import math

import pytest
import torch
import torch.nn.functional as F

pytestmark = pytest.mark.gpu
dev = "cuda"


@pytest.fixture(scope="module")
def NN():
    from simpledepthestimation_amd.hip import nn
    return nn


def nhwc(x):
    return x.permute(0, 2, 3, 1).contiguous().to(torch.bfloat16).to(dev)


def nchw(y, C):
    return y[..., :C].float().permute(0, 3, 1, 2).cpu()


CASES = [
    # name, B, H, W, C0, C1(upcat skip), Cout, k, stride, pad, reflect, bias, act, expected variant (7<BM><BN>)
    ("1x1_64_256_t128128", 2, 128, 256, 64, 0, 256, 1, 1, 0, False, False, 0, 7128128),
    ("1x1_256_64_t128064", 2, 128, 256, 256, 0, 64, 1, 1, 0, False, False, 0, 7128064),
    ("1x1_s2_256_512", 2, 48, 80, 256, 0, 512, 1, 2, 0, False, False, 0, 7064064),
    ("1x1_128_40_raggedN", 3, 21, 37, 128, 0, 40, 1, 1, 0, False, True, 1, 7064064),
    ("3x3_64_64_ragged", 3, 23, 41, 64, 0, 64, 3, 1, 1, False, False, 0, 7064064),
    ("3x3_s2_128_128", 2, 48, 80, 128, 0, 128, 3, 2, 1, False, False, 0, 7064064),
    ("3x3_256_256_splitk", 2, 12, 40, 256, 0, 256, 3, 1, 1, False, False, 0, 7064064),
    ("3x3_512_512_tinyM", 2, 6, 20, 512, 0, 512, 3, 1, 1, False, False, 0, 7064064),
    ("refl_128_64_elu", 2, 24, 40, 128, 0, 64, 3, 1, 1, True, True, 1, 7064064),
    ("refl_64_24_elu", 2, 24, 40, 64, 0, 24, 3, 1, 1, True, True, 1, 7064064),
    ("upcat_64_64_128", 2, 12, 20, 64, 64, 128, 3, 1, 1, True, True, 1, 7064064),
    ("upcat_256_1024_256", 1, 6, 10, 256, 1024, 256, 3, 1, 1, True, True, 1, 7064064),
    ("upcat_64_0_64", 2, 12, 20, 64, 0, 64, 3, 1, 1, True, True, 1, 7064064),
    ("5x5_64_64", 1, 12, 20, 64, 0, 64, 5, 1, 2, False, True, 0, 7064064),
    # stride 2: the data gradient runs as four parity classes of output pixels (dense sub-convolutions over the taps that meet a real pixel)
    ("3x3_s2_64_64_odd", 2, 13, 21, 64, 0, 64, 3, 2, 1, False, False, 0, 7064064),
    ("5x5_s2_64_128", 1, 16, 24, 64, 0, 128, 5, 2, 2, False, True, 0, 7064064),
    ("7x7_s2_64_64", 1, 17, 23, 64, 0, 64, 7, 2, 3, False, False, 0, 7064064),
    ("1x1_s2_64_64_odd", 2, 13, 21, 64, 0, 64, 1, 2, 0, False, False, 0, 7064064),
    ("3x3_64_64_big_t128064", 2, 128, 256, 64, 0, 64, 3, 1, 1, False, False, 0, 7128064),
    ("3x3_128_256_t128128", 2, 24, 40, 128, 0, 256, 3, 1, 1, False, True, 1, 7128128),
    ("upcat_64_64_128_t128064", 2, 12, 20, 64, 64, 128, 3, 1, 1, True, True, 1, 7128064),
]


def ulp_close(a, b, what):
    """bf16 tensors that may differ by summation order only: <= 1 bf16 ulp (2^-8 relative) + a small absolute term."""
    a, b = a.float(), b.float()
    err = (a - b).abs()
    lim = 2.0 ** -7 * torch.maximum(a.abs(), b.abs()) + 2e-3
    bad = (err > lim).sum().item()
    assert bad == 0, f"{what}: {bad} of {a.numel()} elements differ by more than one bf16 ulp (max {err.max().item():.3e})"


@pytest.mark.parametrize("depth", [4, 3])
@pytest.mark.parametrize("case", CASES, ids=[c[0] for c in CASES])
def test_pgemm_conv(NN, case, depth):
    from simpledepthestimation_amd.hip import lib as L
    name, B, H, W, C0, C1, Cout, k, stride, pad, reflect, has_bias, act, variant = case
    g = torch.Generator().manual_seed(len(name) * 7 + B)
    upcat = name.startswith("upcat")
    x0 = torch.randn(B, C0, H, W, generator=g).bfloat16().float()
    x1 = torch.randn(B, C1, 2 * H, 2 * W, generator=g).bfloat16().float() if C1 else None
    Cin = C0 + C1
    w = (torch.randn(Cout, Cin, k, k, generator=g) / math.sqrt(Cin * k * k)).bfloat16().float()
    b = torch.randn(Cout, generator=g) * 0.1 if has_bias else None
    # torch-CPU fp32 reference on the rounded operands
    xr = x0.clone().requires_grad_(True)
    x1r = x1.clone().requires_grad_(True) if C1 else None
    xin = xr
    if upcat:
        up = F.interpolate(xr, scale_factor=2, mode="nearest")
        xin = torch.cat([up, x1r], 1) if C1 else up
    if reflect:
        xin = F.pad(xin, (1, 1, 1, 1), mode="reflect")
    yr = F.conv2d(xin, w, b, stride, 0 if reflect else pad)
    if act == 1:
        yr = F.elu(yr)
    gy = torch.randn(yr.shape, generator=g).bfloat16().float()
    yr.backward(gy)

    old_depth = NN.set_option(NN.OPT_PGEMM_DEPTH, depth)
    old_3x3 = NN.set_option(NN.OPT_PGEMM_3X3, 1)        # also the layers the LDS-halo kernel would take
    old_tile = NN.set_option(NN.OPT_PGEMM_TILE, variant - 7000000)      # the dispatcher's own choice is 64x64 everywhere; the larger tiles are forced here
    try:
        outs = {}
        for on in (1, 0):
            NN.set_option(NN.OPT_PGEMM, on)
            xd = nhwc(x0).requires_grad_(True)
            x1d = nhwc(x1).requires_grad_(True) if C1 else None
            wd = w.clone().to(dev).requires_grad_(True)
            bd = b.clone().to(dev).requires_grad_(True) if has_bias else None
            res = NN.conv2d(xd, wd, bd, stride=stride, pad=pad, reflect=reflect, act=act, skip=x1d, upsample=upcat, bn_stats=not has_bias)
            y, stats = res if isinstance(res, tuple) else (res, None)
            if on:
                d = NN._desc(xd, x1d, NN.SRC_UPCAT if upcat else NN.SRC_PLAIN, k, k, stride, pad, reflect,
                             2 * H if upcat else H, 2 * W if upcat else W, y.shape[1], y.shape[2])
                assert L.lib().sde_conv_fwd_variant(ctypes.byref(d), y.shape[3]) == variant
            y.backward(nhwc(gy) if y.shape[3] == Cout else F.pad(nhwc(gy), (0, y.shape[3] - Cout)))
            tot = None
            if stats is not None:
                tiles = stats.shape[0] - NN.REDUCE_ROWS
                tot = stats[:tiles].double().sum(0).cpu()
            outs[on] = (y.detach().cpu(), tot, xd.grad.detach().cpu(), x1d.grad.detach().cpu() if C1 else None)
    finally:
        NN.set_option(NN.OPT_PGEMM, 1)
        NN.set_option(NN.OPT_PGEMM_DEPTH, old_depth)
        NN.set_option(NN.OPT_PGEMM_3X3, old_3x3)
        NN.set_option(NN.OPT_PGEMM_TILE, old_tile)
    y1, st1, dx1, ds1 = outs[1]
    y0, st0, dx0, ds0 = outs[0]
    if y1.shape[3] > Cout:
        assert (y1[..., Cout:] == 0).all(), "padded output channels must be exact zeros"
    e = ((nchw(y1, Cout).double() - yr.detach().double()).norm() / yr.detach().double().norm()).item()
    assert e < 6e-3, f"y vs fp32 CPU: relative L2 error {e:.3e}"
    ulp_close(y1, y0, "y (pgemm vs register-staged kernel)")
    ulp_close(dx1, dx0, "dX")
    e = ((nchw(dx1, C0).double() - xr.grad.double()).norm() / xr.grad.double().norm()).item()
    assert e < 6e-3, f"dX vs fp32 CPU: relative L2 error {e:.3e}"
    if C1:
        ulp_close(ds1, ds0, "dSkip")
    if st1 is not None:
        yy = y1[..., :Cout].double().reshape(-1, Cout)
        ref = torch.stack([yy.sum(0), (yy * yy).sum(0)], 1)
        assert torch.allclose(st1, ref, rtol=1e-4, atol=1e-3), f"BN partial sums vs sums of the stored outputs: {(st1 - ref).abs().max().item():.3e}"
        assert torch.allclose(st1, st0, rtol=2e-2, atol=2e-1)
