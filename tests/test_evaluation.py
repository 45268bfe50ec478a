"""KITTI evaluator.  CPU part: the numpy oracle against tests/golden/eval.npz -- outputs of the REFERENCE's own garg_crop /
eigen_crop / compute_errors (oracle/gen_golden_eval.py) -- and the product's host logic (crop windows, backward index maps) against the oracle.
GPU part: sde_depth_metrics through the C ABI and the evaluator classes against both.

Tolerances: valid-pixel counts, crop windows, index maps and medians are exact; the nine metrics are float reductions -- the reference reduces in
float32 (numpy pairwise), the kernel in float64 over float32 terms -- compared at 2e-5 relative (abs_rel's north-star tolerance is 2e-3 absolute)."""
import os

import numpy as np
import pytest
import torch

from oracle import evaluation as OE

CASES = ("k0", "k1", "small")
RANGES = ((1e-3, 80), (1e-3, 30), (30, 50), (50, 80))


def _key(tag, crop, scale, lo, hi):
    return f"{tag}.{crop}.s{scale}.{lo:g}_{hi:g}"


# ------------------------------------------------------------------------------------------------ CPU: oracle pinned to the reference
@pytest.mark.parametrize("tag", CASES)
def test_oracle_matches_reference_outputs(evg, tag):
    pred, gt, ymap, xmap = evg[f"{tag}.pred"], evg[f"{tag}.gt"], evg[f"{tag}.ymap"], evg[f"{tag}.xmap"]
    for crop in ("garg", "eigen"):
        y0, y1, x0, x1 = OE.crop_window(crop, *gt.shape)
        assert (y1 - y0, x1 - x0) == tuple(evg[f"{tag}.{crop}.shape"])
        for scale in (0, 1):
            for lo, hi in RANGES:
                want = evg[_key(tag, crop, scale, lo, hi)]
                got = OE.process_image(pred, gt, ymap, xmap, crop, lo, hi, bool(scale))
                if want[9] == 0:
                    assert got is None
                    continue
                np.testing.assert_allclose(np.array(got, np.float64), want[:9], rtol=1e-6, atol=0)


def test_host_maps_and_windows_match_oracle():
    from simpledepthestimation_amd.evaluation import depth_evaluation as DE
    for h, w in ((375, 1242), (370, 1226), (376, 1241), (48, 160)):
        for kind in ("garg", "eigen", None):
            assert DE.crop_window(kind, h, w) == OE.crop_window(kind, h, w)
    meta = {"h_before_resize": 375, "w_before_resize": 1242}
    for got, want in zip(DE.backward_maps((192, 640), meta, ["LoadImg", "Resize", "ToTensor"]), OE.backward_maps((192, 640), meta, ["Resize"])):
        assert got.dtype == np.int32 and np.array_equal(got, want)
    meta = {"h_before_kb_crop": 375, "w_before_kb_crop": 1242, "kb_y_start": 23, "kb_x_start": 13}
    ry, rx = DE.backward_maps((352, 1216), meta, ["KBCrop"])
    oy, ox = OE.backward_maps((352, 1216), meta, ["KBCrop"])
    assert np.array_equal(ry, oy) and np.array_equal(rx, ox)
    assert ry[22] == -1 and ry[23] == 0 and ry[374] == 351 and rx[12] == -1 and rx[13] == 0 and rx[1228] == 1215 and rx[1229] == -1
    # Resize composed with a crop (forward order KBCrop -> Resize): backward = un-resize, then un-crop
    meta.update(h_before_resize=352, w_before_resize=1216)
    ry, rx = DE.backward_maps((192, 640), meta, ["KBCrop", "Resize"])
    oy, ox = OE.backward_maps((192, 640), meta, ["KBCrop", "Resize"])
    assert np.array_equal(ry, oy) and np.array_equal(rx, ox) and ry.size == 375 and rx.size == 1242 and ry.max() == 191 and rx.max() == 639
    # integer up-sampling factors: the cv2 rule reduces to y // factor
    assert np.array_equal(OE.nearest_map(24, 48), np.arange(48) // 2)


def test_registry_and_config_surface():
    from simpledepthestimation_amd.config import get_project_cfg
    from simpledepthestimation_amd.evaluation import EVALUATOR_REGISTRY, DatasetEvaluator, build_evaluator
    cfg = get_project_cfg("MonoDepth2")
    evs = build_evaluator(cfg, None)
    assert [type(e).__name__ for e in evs] == list(cfg.EVALUATORS) and all(isinstance(e, DatasetEvaluator) for e in evs)
    assert [(e.min_depth, e.max_depth, e.tag) for e in evs] == [(1e-3, 80, "kitti evaluator"), (1e-3, 30, "kitti evaluator (0-30m)"),
                                                                (30, 50, "kitti evaluator (30-50m)"), (50, 80, "kitti evaluator (50-80m)")]
    assert evs[0].preprocess_chain == ["Resize"] and evs[0].evaluate() == {}
    with pytest.raises(KeyError):
        EVALUATOR_REGISTRY.get("nope")


# ------------------------------------------------------------------------------------------------ GPU
gpu = pytest.mark.gpu
dev = "cuda"


@gpu
@pytest.mark.parametrize("tag", CASES)
def test_kernel_vs_reference_golden(evg, tag):
    from simpledepthestimation_amd.hip import evaluation as HE
    pred, gt = evg.t(f"{tag}.pred").to(dev), evg.t(f"{tag}.gt").to(dev)
    ymap, xmap = evg.t(f"{tag}.ymap").to(dev), evg.t(f"{tag}.xmap").to(dev)
    for crop in ("garg", "eigen"):
        win = OE.crop_window(crop, *gt.shape)
        for scale in (0, 1):
            for lo, hi in RANGES:
                want = evg[_key(tag, crop, scale, lo, hi)]
                got = HE.depth_metrics(pred, gt, ymap, xmap, win, lo, hi, scale).cpu().numpy()
                assert got[9] == want[9], (crop, scale, lo, hi)
                if want[9] == 0:
                    continue
                np.testing.assert_allclose(got[:9], want[:9], rtol=2e-5, atol=1e-7, err_msg=f"{crop} s{scale} {lo}-{hi}")
                if scale:
                    assert np.array_equal(got[10:12].astype(np.float32), evg[f"{tag}.{crop}.medians"].astype(np.float32))


@gpu
def test_kernel_edge_cases():
    from simpledepthestimation_amd.hip import evaluation as HE
    g = torch.Generator().manual_seed(5)
    pred = (torch.rand(20, 30, generator=g) * 50 + 1).to(dev)
    ymap, xmap = torch.arange(20, dtype=torch.int32, device=dev), torch.arange(30, dtype=torch.int32, device=dev)
    # no valid pixel at all: count 0 (the evaluator skips the image), nothing non-finite is required of the rest
    out = HE.depth_metrics(pred, torch.zeros(20, 30, device=dev), ymap, xmap, (0, 20, 0, 30), 1e-3, 80, 1).cpu()
    assert out[9] == 0
    # one and two valid pixels: median of one element / mean of two; pixels outside the prediction (-1) read 0 -> excluded by nothing, as in the reference
    gt = torch.zeros(20, 30); gt[3, 4] = 10.0; gt[7, 8] = 20.0
    out = HE.depth_metrics(pred, gt.to(dev), ymap, xmap, (0, 20, 0, 30), 1e-3, 80, 1).cpu().numpy()
    p = pred.cpu().numpy(); want = OE.process_image(p, gt.numpy(), ymap.cpu().numpy(), xmap.cpu().numpy(), None, 1e-3, 80, True)
    assert out[9] == 2 and out[10] == 15.0 and np.float32(out[11]) == np.float32((p[3, 4] + p[7, 8]) * np.float32(0.5))
    np.testing.assert_allclose(out[:9], np.array(want, np.float64), rtol=2e-5, atol=1e-7)
    # window arguments are validated on the host side of the ABI
    with pytest.raises(Exception):
        HE.depth_metrics(pred, gt.to(dev), ymap, xmap, (0, 21, 0, 30), 1e-3, 80, 0)
    with pytest.raises(Exception):
        HE.depth_metrics(pred, gt.to(dev), ymap[:5], xmap, (0, 20, 0, 30), 1e-3, 80, 0)


@gpu
@pytest.mark.parametrize("gt_scale", [False, True])
def test_evaluators_end_to_end_vs_oracle(evg, gt_scale):
    """process() over two batches (KITTI-sized images of different sizes + one image without valid ground truth) == the oracle's loop."""
    from simpledepthestimation_amd.config import get_project_cfg
    from simpledepthestimation_amd.evaluation import DatasetEvaluators, build_evaluator
    cfg = get_project_cfg("MonoDepth2"); cfg.TEST.GT_SCALE = gt_scale
    evs = build_evaluator(cfg, None)
    both = DatasetEvaluators(evs)
    both.reset()
    images = [(evg[f"{t}.pred"], evg[f"{t}.gt"]) for t in ("k0", "k1")] + [(evg["k0.pred"], np.zeros((375, 1242), np.float32))]
    metas = [{"h_before_resize": g.shape[0], "w_before_resize": g.shape[1]} for _, g in images]
    for lo in (0, 2):
        sel = slice(lo, lo + 2)
        inputs = {"depth_orig": [g[None] for _, g in images[sel]], "metadata": metas[sel]}
        outputs = {"depth_pred": torch.stack([torch.from_numpy(p) for p, _ in images[sel]]).unsqueeze(1).to(dev)}
        both.process(inputs, outputs)
    res = both.evaluate()
    assert list(res) == [e.tag for e in evs]
    for ev in evs:
        rows = []
        for (p, g), m in zip(images, metas):
            ymap, xmap = OE.backward_maps(p.shape, m, ["Resize"])
            r = OE.process_image(p, g, ymap, xmap, "garg", ev.min_depth, ev.max_depth, gt_scale)
            if r is not None:
                rows.append(r)
        want = np.mean(np.array(rows, np.float64), axis=0)
        names = ("abs_rel", "sq_rel", "rms", "log_rms", "d1", "d2", "d3")
        assert list(res[ev.tag]) == list(names)
        np.testing.assert_allclose([res[ev.tag][n] for n in names], want[2:], rtol=2e-5, atol=1e-7, err_msg=ev.tag)


@gpu
def test_inference_on_dataset_runs_the_model_in_eval_mode():
    from oracle import models as OM
    from oracle.gen_golden import sup_batch
    from simpledepthestimation_amd.config import get_project_cfg
    from simpledepthestimation_amd.evaluation import build_evaluator, inference_on_dataset
    from simpledepthestimation_amd.modeling import build_model
    cfg = get_project_cfg("MonoDepth2"); cfg.MODEL.META_ARCHITECTURE = "SupDepthModel"; cfg.MODEL.DEPTH_NET.ENCODER_NAME = "18"; cfg.MODEL.DEVICE = dev
    cfg.TEST.GT_SCALE = False
    model = build_model(cfg); model.load_state_dict(OM.init_state_dict(18, seed=31), strict=True); model.train()
    g = np.random.default_rng(3)
    loader = []
    for i in range(2):
        b = sup_batch(2, 64, 192, 40 + i)
        gts = [np.where(g.random((1, 128, 384)) < 0.3, g.random((1, 128, 384)) * 60 + 2, 0).astype(np.float32) for _ in range(2)]
        loader.append({"img": b["img"], "depth_orig": gts, "metadata": [{"h_before_resize": 128, "w_before_resize": 384}] * 2})
    res = inference_on_dataset(model, loader, build_evaluator(cfg, None))
    assert model.training                                   # restored
    with torch.no_grad():
        model.eval()
        preds = [model({"img": b["img"]})["depth_pred"].cpu().numpy() for b in loader]
    rows = []
    for b, pp in zip(loader, preds):
        for gt, p in zip(b["depth_orig"], pp):
            ymap, xmap = OE.backward_maps(p.shape[-2:], b["metadata"][0], ["Resize"])
            rows.append(OE.process_image(p.squeeze(), gt.squeeze(), ymap, xmap, "garg", 1e-3, 80, False))
    want = np.mean(np.array(rows, np.float64), axis=0)
    got = res["kitti evaluator"]
    np.testing.assert_allclose([got[n] for n in ("abs_rel", "sq_rel", "rms", "log_rms", "d1", "d2", "d3")], want[2:], rtol=2e-5, atol=1e-7)


def test_depth_saver_writes_the_reference_png_format(tmp_path):
    """CPU tensors suffice here: the saver's restore is an index gather.  16-bit PNG of depth * 255, truncated (file_utils.py:L5-8), named
    <date>_<drive>_<img_id>.png, restored to the original size through the Resize index maps."""
    from PIL import Image
    from simpledepthestimation_amd.config import get_project_cfg
    from simpledepthestimation_amd.evaluation import EVALUATOR_REGISTRY
    rng = np.random.default_rng(0)
    pred = (rng.random((2, 1, 24, 80)) * 80).astype(np.float32)
    metas = [{"date": "2011_09_26", "drive": "0002", "img_id": f"{i:010d}", "h_before_resize": 37, "w_before_resize": 124} for i in range(2)]
    saver = EVALUATOR_REGISTRY.get("kitti_depth_saver")(get_project_cfg("MonoDepth2"), str(tmp_path / "out"))
    saver.process({"metadata": metas}, {"depth_pred": torch.from_numpy(pred)})
    assert saver.evaluate() is None
    assert sorted(os.listdir(tmp_path / "out")) == ["2011_09_26_0002_0000000000.png", "2011_09_26_0002_0000000001.png"]
    for i in range(2):
        img = Image.open(tmp_path / "out" / f"2011_09_26_0002_{i:010d}.png")
        got = np.array(img)
        ymap, xmap = OE.backward_maps((24, 80), metas[i], ["Resize"])
        want = (pred[i, 0][ymap[:, None], xmap[None, :]] * 255).astype(np.uint16)
        assert got.dtype == np.uint16 and got.shape == (37, 124) and np.array_equal(got, want)
