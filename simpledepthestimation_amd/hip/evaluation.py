"""ctypes binding of sde_depth_metrics (include/sde_hip.h, csrc/eval.hip): the KITTI depth metrics of one image, computed on the GPU."""
from ctypes import POINTER, c_double, c_int, c_void_p

import torch

from . import lib as L

_P, _I, _F = c_void_p, c_int, L.c_float
L.register_protos({
    "sde_depth_metrics_num_blocks": ([_I, _I], c_int),
    "sde_depth_metrics": ([_P, _I, _I, _P, _I, _I, _P, _P, _I, _I, _I, _I, _F, _F, _I, _P, _P, _P, _P, _P], c_int),
})
NSUM = 11          # SDE_EVAL_NSUM
NOUT = 12


def depth_metrics(pred, gt, ymap, xmap, window, min_depth, max_depth, gt_scale, med=None):
    """pred [ph,pw] fp32, gt [gh,gw] fp32, ymap [gh] / xmap [gw] int32 (CUDA tensors); window = (y0, y1, x0, x1) in gt coordinates.
    Returns a CUDA float64 tensor [12]: the nine compute_errors values, the valid-pixel count, median(gt), median(pred).  No host sync.
    `med`: the 4-float workspace of an earlier call on the same image and window -- its medians are reused instead of selected again."""
    if pred.dtype != torch.float32 or gt.dtype != torch.float32 or ymap.dtype != torch.int32 or xmap.dtype != torch.int32:
        raise L.SdeHipError("depth_metrics: pred / gt must be float32 and the index maps int32")
    if pred.dim() != 2 or gt.dim() != 2 or ymap.numel() != gt.shape[0] or xmap.numel() != gt.shape[1]:
        raise L.SdeHipError(f"depth_metrics: shapes pred {tuple(pred.shape)} gt {tuple(gt.shape)} maps {ymap.numel()} x {xmap.numel()}")
    y0, y1, x0, x1 = (int(v) for v in window)
    lib = L.lib()
    dev = gt.device
    part = torch.empty(lib.sde_depth_metrics_num_blocks(y1 - y0, x1 - x0) * NSUM, device=dev, dtype=torch.float64)
    mode = (2 if med is not None else 1) if gt_scale else 0
    keys = torch.empty(2 * (y1 - y0) * (x1 - x0), device=dev, dtype=torch.int32) if mode == 1 else None
    if med is None:
        med = torch.zeros(4, device=dev, dtype=torch.float32)       # medians, key count (float), key counter (uint32): zeroed
    out = torch.empty(NOUT, device=dev, dtype=torch.float64)
    L.check(lib.sde_depth_metrics(L.ptr(pred), pred.shape[0], pred.shape[1], L.ptr(gt), gt.shape[0], gt.shape[1], L.ptr(ymap), L.ptr(xmap), y0, y1, x0, x1,
                                  float(min_depth), float(max_depth), mode, L.ptr(part), L.ptr(med), L.ptr(keys), L.ptr(out), L.stream()),
            "sde_depth_metrics")
    out.med = med
    return out
