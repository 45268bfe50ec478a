"""KITTI depth evaluators on the GPU (contract: detectron2/evaluation/depth_evaluation.py:L56-160).

``kitti_evaluator`` and its three range variants keep the reference's names, constructor ``(cfg, output_folder)``, ``process(inputs, outputs)``
inputs (``inputs['depth_orig']`` ground truth at original resolution, ``inputs['metadata']``, ``outputs['depth_pred']`` [B,1,h,w]) and the
``{tag: {abs_rel, sq_rel, rms, log_rms, d1, d2, d3}}`` result.  Per image one ``sde_depth_metrics`` call replaces the numpy body of the loop:
the nearest-neighbour resize back to the original size (and the crop un-pastes) are two index maps the kernel reads the network output
through, the Garg crop is a window, the optional median scaling and the nine error sums run on the device, and nothing is copied to the host
until ``evaluate()``.  The four evaluators of a config share the uploaded ground truth and the medians of an image through the ``outputs`` dict.

``kitti_depth_saver`` writes the restored full-size predictions as 16-bit PNGs with the reference's scaling (``depth * 255`` truncated to uint16,
utils/file_utils.py:L5-8: the same "/255" convention its loader reads back, loading.py:L59) through Pillow instead of cv2."""
import logging
import os

import numpy as np
import torch
import torch.distributed as dist

from ..hip import evaluation as HE
from .evaluator import EVALUATOR_REGISTRY, DatasetEvaluator

METRIC_NAMES = ("silog", "log10", "abs_rel", "sq_rel", "rms", "log_rms", "d1", "d2", "d3")


def crop_window(kind, h, w):
    """garg_crop / eigen_crop (L16-27) as the window [y0, y1) x [x0, x1) they slice out of an h x w map."""
    fy0, fy1, fx0, fx1 = {"garg": (0.40810811, 0.99189189, 0.03594771, 0.96405229),
                          "eigen": (0.3324324, 0.91351351, 0.0359477, 0.96405229), None: (0.0, 1.0, 0.0, 1.0)}[kind]
    return int(fy0 * h), int(fy1 * h), int(fx0 * w), int(fx1 * w)


def _nearest(src, dst):
    """Source index per destination index of cv2.resize(INTER_NEAREST): floor(x * (1 / (dst/src))) in doubles, clamped (OpenCV resizeNN)."""
    step = 1.0 / (float(dst) / float(src))
    return np.minimum(np.floor(np.arange(dst, dtype=np.float64) * step).astype(np.int64), src - 1).astype(np.int32)


def _as_int(v):
    """Integer-like metadata value (python / numpy int, 0-d integer tensor or array) as a python int, else None."""
    if isinstance(v, (bool, np.bool_)):
        return None
    if isinstance(v, (int, np.integer)):
        return int(v)
    if torch.is_tensor(v) and v.ndim == 0 and not v.is_floating_point() and not v.is_complex():
        return int(v.item())
    if isinstance(v, np.ndarray) and v.ndim == 0 and np.issubdtype(v.dtype, np.integer):
        return int(v)
    return None


def backward_maps(pred_hw, metadata, chain):
    """The postprocess.backward() chain (augmentation.py: Resize L163-166, KBCrop L67-74, CropTopTo L113-120) as (ymap, xmap) int32 arrays:
    which network-output row / column each row / column of the restored full-size map shows; -1 where the reference pastes zeros."""
    rows, cols = np.arange(pred_hw[0], dtype=np.int32), np.arange(pred_hw[1], dtype=np.int32)
    for step in reversed(list(chain)):
        if step == "Resize":
            rows = rows[_nearest(rows.size, int(metadata["h_before_resize"]))]
            cols = cols[_nearest(cols.size, int(metadata["w_before_resize"]))]
        elif step == "KBCrop":
            big_r = np.full(int(metadata["h_before_kb_crop"]), -1, np.int32)
            big_c = np.full(int(metadata["w_before_kb_crop"]), -1, np.int32)
            oy, ox = int(metadata["kb_y_start"]), int(metadata["kb_x_start"])
            big_r[oy:oy + rows.size], big_c[ox:ox + cols.size] = rows, cols
            rows, cols = big_r, big_c
        elif step == "CropTopTo":
            big_r = np.full(int(metadata["h_before_crop"]), -1, np.int32)
            big_r[int(metadata["crop_y_start"]):] = rows
            rows = big_r
    return rows, cols


@EVALUATOR_REGISTRY.register()
class kitti_evaluator(DatasetEvaluator):
    def __init__(self, cfg, output_folder=None):
        super().__init__(cfg)
        self._logger = logging.getLogger(__name__)
        self.min_depth, self.max_depth = 1e-3, 80
        self.garg_crop, self.eigen_crop = True, False
        self.use_gt_scale = bool(cfg.TEST.GT_SCALE)
        self.tag = "kitti evaluator"
        self.metrics = []
        self._maps = {}

    def reset(self):
        self.metrics = []

    def _device_maps(self, pred_hw, gt_hw, meta, device):
        chain = self.preprocess_chain
        key = (pred_hw, gt_hw, tuple(chain), tuple(sorted((k, _as_int(v)) for k, v in meta.items() if _as_int(v) is not None)))
        hit = self._maps.get(key)
        if hit is None:
            if chain:
                rows, cols = backward_maps(pred_hw, meta, chain)
            else:                                   # no chain configured: nearest resize straight to the ground-truth size
                rows, cols = _nearest(pred_hw[0], gt_hw[0]), _nearest(pred_hw[1], gt_hw[1])
            if (rows.size, cols.size) != tuple(gt_hw):
                raise ValueError(f"restored prediction is {rows.size}x{cols.size}, ground truth {gt_hw[0]}x{gt_hw[1]}")
            hit = (torch.from_numpy(rows).to(device), torch.from_numpy(cols).to(device))
            self._maps[key] = hit
        return hit

    def process(self, inputs, outputs):
        preds = outputs["depth_pred"]
        metas = inputs.get("metadata") or [{}] * len(preds)
        # the evaluators of a config see the same batch: ground truth is uploaded once, and the medians of GT scaling (same 1e-3..80 mask,
        # same window) are selected once -- both are parked in the outputs dict for the evaluators that run after this one
        shared = outputs.setdefault("_sde_eval_shared", {})
        kind = "garg" if self.garg_crop else ("eigen" if self.eigen_crop else None)
        for i, (gt, pred, meta) in enumerate(zip(inputs["depth_orig"], preds, metas)):
            pred = pred.detach().squeeze().float().contiguous()
            gt_dev = shared.get(("gt", i))
            if gt_dev is None:
                gt_dev = torch.as_tensor(np.ascontiguousarray(gt) if isinstance(gt, np.ndarray) else gt).squeeze().to(pred.device, torch.float32).contiguous()
                shared["gt", i] = gt_dev
            rows, cols = self._device_maps(tuple(pred.shape), tuple(gt_dev.shape), meta, pred.device)
            res = HE.depth_metrics(pred, gt_dev, rows, cols, crop_window(kind, *gt_dev.shape), self.min_depth, self.max_depth, self.use_gt_scale,
                                   med=shared.get(("med", i, kind)) if self.use_gt_scale else None)
            if self.use_gt_scale:
                shared["med", i, kind] = res.med
            self.metrics.append(res)

    def evaluate(self):
        rows = torch.stack(self.metrics).cpu().numpy() if self.metrics else np.zeros((0, HE.NOUT))      # the run's only device -> host copy
        rows = rows[rows[:, 9] > 0][:, :9]                                                            # images without a valid pixel are skipped (L102)
        if dist.is_available() and dist.is_initialized() and dist.get_world_size() > 1:
            gathered = [None] * dist.get_world_size()
            dist.all_gather_object(gathered, rows)
            if dist.get_rank() != 0:
                return {}
            rows = np.concatenate(gathered, 0)
        if rows.shape[0] == 0:
            self._logger.warning("[DepthEvaluator] Did not receive valid predictions.")
            return {}
        mean = rows.mean(0)
        self._logger.info("%s%s: %s", self.tag, " w/ gt scale" if self.use_gt_scale else "",
                          ", ".join(f"{n} {mean[i]:.3f}" for i, n in enumerate(METRIC_NAMES) if i >= 2))
        return {self.tag: {n: float(mean[i]) for i, n in enumerate(METRIC_NAMES) if i >= 2}}


def _ranged(name, lo, hi, tag):
    def __init__(self, cfg, output_folder=None):
        kitti_evaluator.__init__(self, cfg, output_folder)
        self.min_depth, self.max_depth, self.tag = lo, hi, tag
    return EVALUATOR_REGISTRY.register(type(name, (kitti_evaluator,), {"__init__": __init__, "__doc__": f"kitti_evaluator restricted to {lo} < gt < {hi}"}))


kitti_evaluator_0_30 = _ranged("kitti_evaluator_0_30", 1e-3, 30, "kitti evaluator (0-30m)")
kitti_evaluator_30_50 = _ranged("kitti_evaluator_30_50", 30, 50, "kitti evaluator (30-50m)")
kitti_evaluator_50_80 = _ranged("kitti_evaluator_50_80", 50, 80, "kitti evaluator (50-80m)")


def write_depth(depth, save_path):
    """utils/file_utils.py:L5-8: uint16 PNG of ``depth * 255`` (truncation, as ``astype(np.uint16)``)."""
    from PIL import Image
    scaled = (np.asarray(depth, dtype=np.float32) * 255).astype(np.uint16)
    Image.fromarray(scaled).save(save_path, format="PNG", compress_level=3)


def np_median(t):
    """np.median of a 1-D tensor (depth_evaluation.py:L96-97 uses numpy): the MEAN of the two middle elements for an even count --
    torch.Tensor.median() would return the lower one."""
    n = t.numel()
    if n == 0:
        return t.new_tensor(float("nan"))
    s = torch.sort(t.reshape(-1).float())[0]
    return (s[(n - 1) // 2] + s[n // 2]) * 0.5


@EVALUATOR_REGISTRY.register()
class kitti_depth_saver(DatasetEvaluator):
    """Saves every prediction, restored to the original image size by the same index maps the metric evaluators use (depth_evaluation.py:L163-203),
    to ``<output_folder>/<date>_<drive>_<img_id>.png``.  The gather through the maps runs on the device; one device -> host copy per image."""

    def __init__(self, cfg, output_folder):
        super().__init__(cfg)
        self._logger = logging.getLogger(__name__)
        self.use_gt_scale = bool(cfg.TEST.GT_SCALE)
        self.output_folder = output_folder
        self._maps = {}

    def process(self, inputs, outputs):
        for i, (pred, meta) in enumerate(zip(outputs["depth_pred"], inputs["metadata"])):
            pred = pred.detach().squeeze().float()
            if self.preprocess_chain:
                rows, cols = backward_maps(tuple(pred.shape), meta, self.preprocess_chain)
                ry, cx = torch.from_numpy(rows).to(pred.device).long(), torch.from_numpy(cols).to(pred.device).long()
                full = pred[ry.clamp(min=0)][:, cx.clamp(min=0)] * ((ry >= 0)[:, None] & (cx >= 0)[None, :])
            else:
                full = pred
            if self.use_gt_scale and "depth_gt_orig" in inputs:
                gt = torch.as_tensor(inputs["depth_gt_orig"][i]).squeeze().to(full.device, torch.float32)
                valid = (gt > 1e-3) & (gt < 80)
                full = full * np_median(gt[valid]) / np_median(full[valid])
            name = f"{meta['date']}_{meta['drive']}_{meta['img_id']}.png"
            os.makedirs(self.output_folder, exist_ok=True)
            write_depth(full.cpu().numpy(), os.path.join(self.output_folder, name))

    def evaluate(self):
        self._logger.info("depth saved to %s%s", self.output_folder, " w/ gt scale" if self.use_gt_scale else "")
        return None
