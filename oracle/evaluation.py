"""TEST INFRASTRUCTURE -- CPU restatement (numpy) of the reference's KITTI evaluator arithmetic; never imported by the product path.

Follows detectron2/evaluation/depth_evaluation.py: garg_crop L16-20, eigen_crop L23-27, compute_errors L30-53, kitti_evaluator.process L74-104
and evaluate L106-135, plus the postprocess.backward() chain of detectron2/data/preprocess/augmentation.py (KBCrop L67-74, CropTopTo L113-120,
Resize L163-166).  Pinned by tests/golden/eval.npz, which oracle/gen_golden_eval.py produced by running the reference's own garg_crop /
eigen_crop / compute_errors on seeded inputs.  Resize.backward calls cv2.resize(INTER_NEAREST); cv2 is a third-party dependency that is
neither under /root/reference nor installed here, so `nearest_map` restates OpenCV's published resizeNN index rule
(sx = min(floor(x * (1 / (dst / src))), src - 1), doubles) -- parity unpinned at that one boundary."""
import numpy as np


def crop_window(kind, h, w):
    """[y0, y1) x [x0, x1) of garg_crop / eigen_crop for an h x w ground truth (int() truncation as in the reference)."""
    if kind == "garg":
        return int(0.40810811 * h), int(0.99189189 * h), int(0.03594771 * w), int(0.96405229 * w)
    if kind == "eigen":
        return int(0.3324324 * h), int(0.91351351 * h), int(0.0359477 * w), int(0.96405229 * w)
    return 0, h, 0, w


def compute_errors(gt, pred):
    """float32 arrays of the valid pixels -> (silog, log10, abs_rel, sq_rel, rms, log_rms, d1, d2, d3)."""
    ratio = np.maximum(gt / pred, pred / gt)
    d = [(ratio < 1.25 ** k).mean() for k in (1, 2, 3)]
    diff = gt - pred
    rms = np.sqrt((diff ** 2).mean())
    log_rms = np.sqrt(((np.log(gt) - np.log(pred)) ** 2).mean())
    abs_rel = (np.abs(diff) / gt).mean()
    sq_rel = ((diff ** 2) / gt).mean()
    e = np.log(pred) - np.log(gt)
    silog = np.sqrt((e ** 2).mean() - e.mean() ** 2 + 1e-8) * 100
    log10 = np.abs(np.log10(pred) - np.log10(gt)).mean()
    return silog, log10, abs_rel, sq_rel, rms, log_rms, d[0], d[1], d[2]


def nearest_map(src, dst):
    """index of the source row/column every destination row/column reads under cv2.resize(..., INTER_NEAREST)."""
    inv_scale = float(dst) / float(src)
    step = 1.0 / inv_scale
    return np.minimum(np.floor(np.arange(dst, dtype=np.float64) * step).astype(np.int64), src - 1).astype(np.int32)


def backward_maps(pred_shape, metadata, chain):
    """Compose the postprocess.backward() steps (`chain` = preprocess names in FORWARD order, as in cfg.DATASETS.TEST.PREPROCESS) into
    (ymap, xmap): for every row / column of the final full-size map, the row / column of the network output it shows, -1 = zero fill."""
    ph, pw = pred_shape
    ymap, xmap = np.arange(ph, dtype=np.int32), np.arange(pw, dtype=np.int32)
    for name in reversed(list(chain)):
        if name == "Resize":
            H, W = metadata["h_before_resize"], metadata["w_before_resize"]
            ymap, xmap = ymap[nearest_map(len(ymap), H)], xmap[nearest_map(len(xmap), W)]
        elif name == "KBCrop":
            H, W, y0, x0 = metadata["h_before_kb_crop"], metadata["w_before_kb_crop"], metadata["kb_y_start"], metadata["kb_x_start"]
            ny, nx = np.full(H, -1, np.int32), np.full(W, -1, np.int32)
            ny[y0:y0 + len(ymap)] = ymap; nx[x0:x0 + len(xmap)] = xmap
            ymap, xmap = ny, nx
        elif name == "CropTopTo":
            H, W, y0 = metadata["h_before_crop"], metadata["w_before_crop"], metadata["crop_y_start"]
            ny = np.full(H, -1, np.int32)
            ny[y0:] = ymap
            ymap = ny
            assert len(xmap) == W
    return ymap, xmap


def process_image(pred, gt, ymap, xmap, crop="garg", min_depth=1e-3, max_depth=80.0, gt_scale=False):
    """One iteration of kitti_evaluator.process's loop -> the 9-tuple, or None when no pixel is valid."""
    yy, xx = np.asarray(ymap), np.asarray(xmap)
    full = np.where((yy[:, None] >= 0) & (xx[None, :] >= 0), pred[np.clip(yy, 0, None)[:, None], np.clip(xx, 0, None)[None, :]], np.float32(0))
    full = full.astype(np.float32)
    y0, y1, x0, x1 = crop_window(crop, *gt.shape[:2])
    p, g = full[y0:y1, x0:x1], gt[y0:y1, x0:x1]
    valid = np.logical_and(g > 1e-3, g < 80)
    if gt_scale:
        p = p * np.median(g[valid]) / np.median(p[valid])
    valid = np.logical_and(g > min_depth, g < max_depth)
    if valid.sum() == 0:
        return None
    return compute_errors(g[valid], p[valid])
