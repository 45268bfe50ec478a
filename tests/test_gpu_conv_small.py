"""GPU parity of the narrow-input 3x3 halo kernel (csrc/conv_halo_small.hip) through the C ABI.

The decoder's high-resolution layers with 8 / 16 / 32 input channels (depth_decoder.py:L21-53, L95-110) -- forward (bias + ELU), and the data
gradients of the 16- / 32-output-channel layers, which run as pad-2 correlations of dz with the flipped operand -- are compared with
  * plain torch-CPU fp32 convolution on the 16-bit-rounded operands (tolerance = 16-bit output rounding), and
  * the generic kernels on the same device buffers (SDE_OPT_CONV_SMALL = 0): fp32 accumulation both, so at most one 16-bit ulp apart.
Covers reflect and zero padding, the up-sampled (+ concatenated) source, ragged tiles, Cout = 1 (ldy = 8: the disparity heads) and fp16.
"""
import ctypes
import math

import pytest
import torch
import torch.nn.functional as F

pytestmark = pytest.mark.gpu
dev = "cuda"

CASES = [
    # name, B, H, W (conv input = output size), C0, C1, Cout, reflect, upcat, bias+elu, dtype, dgrad also narrow?
    ("plain32_16", 2, 96, 96, 32, 0, 16, True, False, True, torch.bfloat16),
    ("up16_16", 2, 96, 128, 16, 0, 16, True, True, True, torch.bfloat16),
    ("disp16_1", 2, 96, 96, 16, 0, 1, True, False, True, torch.bfloat16),
    ("disp32_1_ragged", 3, 77, 75, 32, 0, 1, True, False, True, torch.bfloat16),
    ("zero16_32", 2, 90, 100, 16, 0, 32, False, False, False, torch.bfloat16),
    ("zero32_24_ragged", 2, 101, 91, 32, 0, 24, False, False, True, torch.bfloat16),
    ("up16_16_32", 2, 96, 96, 16, 16, 32, True, True, True, torch.bfloat16),
    ("wide64_32_dgrad_only", 2, 96, 96, 64, 0, 32, True, False, True, torch.bfloat16),      # forward: other kernels; data gradient: 32 -> 64 channels
    ("up32_64_32_dgrad_only", 2, 96, 96, 32, 64, 32, True, True, True, torch.bfloat16),     # data gradient: 32 -> 96 channels
    ("plain32_16_fp16", 2, 96, 96, 32, 0, 16, True, False, True, torch.float16),
]


def close16(a, b, dt, what):
    """16-bit tensors that may differ by summation order only: <= 1 ulp + a small absolute term."""
    a, b = a.float(), b.float()
    ulp = 2.0 ** -7 if dt == torch.bfloat16 else 2.0 ** -10
    err = (a - b).abs()
    bad = (err > ulp * torch.maximum(a.abs(), b.abs()) + 2e-3).sum().item()
    assert bad == 0, f"{what}: {bad} of {a.numel()} elements differ by more than one ulp (max {err.max().item():.3e})"


@pytest.mark.parametrize("case", CASES, ids=[c[0] for c in CASES])
def test_conv_small(case):
    from simpledepthestimation_amd.hip import lib as L
    from simpledepthestimation_amd.hip import nn as NN
    name, B, H, W, C0, C1, Cout, reflect, upcat, with_bias, dt = case
    g = torch.Generator().manual_seed(len(name) * 17 + B)
    h0, w0 = (H // 2, W // 2) if upcat else (H, W)
    x0 = torch.randn(B, C0, h0, w0, generator=g).to(dt).float()
    x1 = torch.randn(B, C1, H, W, generator=g).to(dt).float() if C1 else None
    Cin = C0 + C1
    wt = (torch.randn(Cout, Cin, 3, 3, generator=g) / math.sqrt(Cin * 9)).to(dt).float()
    bias = torch.randn(Cout, generator=g) * 0.1 if with_bias else None
    xr = x0.clone().requires_grad_(True)
    x1r = x1.clone().requires_grad_(True) if C1 else None
    xin = xr
    if upcat:
        up = F.interpolate(xr, scale_factor=2, mode="nearest")
        xin = torch.cat([up, x1r], 1) if C1 else up
    if reflect:
        xin = F.pad(xin, (1, 1, 1, 1), mode="reflect")
    yr = F.conv2d(xin, wt, bias, 1, 0 if reflect else 1)
    if with_bias:
        yr = F.elu(yr)
    gy = torch.randn(yr.shape, generator=g).to(dt).float()
    yr.backward(gy)

    def nhwc(t):
        return t.permute(0, 2, 3, 1).contiguous().to(dt).to(dev)

    res, variants = {}, {}
    for on in (1, 0):
        old = NN.set_option(NN.OPT_CONV_SMALL, on)
        try:
            xd = nhwc(x0).requires_grad_(True)
            x1d = nhwc(x1).requires_grad_(True) if C1 else None
            wd = wt.clone().to(dev).requires_grad_(True)
            bd = bias.clone().to(dev).requires_grad_(True) if with_bias else None
            y = NN.conv2d(xd, wd, bd, stride=1, pad=1, reflect=reflect, act=NN.ACT_ELU if with_bias else NN.ACT_NONE, skip=x1d, upsample=upcat)
            d = NN._desc(xd, x1d, NN.SRC_UPCAT if upcat else NN.SRC_PLAIN, 3, 3, 1, 1, reflect, H, W, H, W)
            variants[on] = L.lib().sde_conv_fwd_variant(ctypes.byref(d), y.shape[3])
            gyd = nhwc(gy)
            y.backward(gyd if y.shape[3] == Cout else F.pad(gyd, (0, y.shape[3] - Cout)))
            torch.cuda.synchronize()
            res[on] = (y.detach().cpu(), xd.grad.detach().cpu(), x1d.grad.detach().cpu() if C1 else None)
        finally:
            NN.set_option(NN.OPT_CONV_SMALL, old)
    y1, dx1, ds1 = res[1]
    y0, dx0, ds0 = res[0]
    if Cin <= 32:
        assert variants[1] // 1000000 == 5 and variants[0] // 1000000 != 5, variants
    else:
        assert variants[1] == variants[0]
    if y1.shape[3] > Cout:
        assert (y1[..., Cout:] == 0).all(), "padded output channels must be exact zeros"
    lim = 6e-3 if dt == torch.bfloat16 else 1e-3

    def rel(a, b):
        return ((a.double() - b.double()).norm() / b.double().norm()).item()

    e = rel(y1[..., :Cout].float().permute(0, 3, 1, 2), yr.detach())
    assert e < lim, f"y vs fp32 CPU: relative L2 error {e:.3e}"
    close16(y1, y0, dt, "y (narrow-input halo kernel vs generic)")
    e = rel(dx1.float().permute(0, 3, 1, 2), xr.grad)
    assert e < 1.5 * lim, f"dX vs fp32 CPU: relative L2 error {e:.3e}"
    close16(dx1, dx0, dt, "dX")
    if C1:
        e = rel(ds1.float().permute(0, 3, 1, 2), x1r.grad)
        assert e < 1.5 * lim, f"dSkip vs fp32 CPU: relative L2 error {e:.3e}"
        close16(ds1, ds0, dt, "dSkip")
