"""CPU: the KITTI data path (simpledepthestimation_amd.data; reference detectron2/data/datasets/kitti_v2.py, preprocess/*.py, build.py).

Pinned by tests/golden/data.npz = outputs of the reference's OWN class bodies (KBCrop, CropTopTo, RandomCrop, RandomFlip, ClipDepth,
resize_depth, KittiDepthV2 incl. its collate) on seeded inputs and a synthetic KITTI tree (oracle/gen_golden_data.py rebuilds both).
Parity unpinned, by necessity (cv2 / torchvision are absent from this image and from the reference checkout): Resize's bilinear image
resampling and RandomImageAug's colour arithmetic -- those are held to properties (identity, constants, exact 2x cases, determinism)."""
import copy
import json
import os
import random

import numpy as np
import pytest
import torch

from oracle.gen_golden_data import Cfg, dataset_cfg, make_kitti_tree, sample
from simpledepthestimation_amd.data import DATASET_REGISTRY, DevicePrefetcher, InferenceSampler, build_detection_test_loader, build_detection_train_loader
from simpledepthestimation_amd.data.preprocess import PREPROCESS_REGISTRY, build_preprocess
from simpledepthestimation_amd.data.preprocess import augmentation as A
from conftest import GOLDEN, _Golden


@pytest.fixture(scope="module")
def gd():
    return _Golden(os.path.join(GOLDEN, "data.npz"))


@pytest.fixture(scope="module")
def tree(tmp_path_factory):
    root = tmp_path_factory.mktemp("kitti")
    return str(root), make_kitti_tree(str(root))


def _clone(d):
    return {k: (v.copy() if isinstance(v, np.ndarray) else ([a.copy() for a in v] if isinstance(v, list) else dict(v))) for k, v in d.items()}


@pytest.mark.parametrize("tag,H,W,seed", [("kitti", 375, 1242, 11), ("small", 370, 1226, 12)])
def test_crops_flip_clip_resize_depth_vs_reference(gd, tag, H, W, seed):
    d = sample(seed, H, W, with_mask=True)
    kb = build_preprocess({"NAME": "KBCrop"})
    e = kb.forward(_clone(d))
    assert int(e["img"].astype(np.int64).sum()) == int(gd[f"{tag}.kb.img_sum"]) and np.array_equal(e["img"][:2, :2], gd[f"{tag}.kb.img_corner"])
    assert np.array_equal(e["depth"][::16, ::32], gd[f"{tag}.kb.depth"]) and np.array_equal(e["intrinsics"], gd[f"{tag}.kb.K"])
    assert float(e["mask"].sum()) == float(gd[f"{tag}.kb.mask_sum"]) and sum(int(a.astype(np.int64).sum()) for a in e["ctx_img"]) == int(gd[f"{tag}.kb.ctx_sum"])
    assert [e["metadata"][k] for k in ("kb_y_start", "kb_x_start", "h_before_kb_crop", "w_before_kb_crop")] == list(gd[f"{tag}.kb.meta"])
    pred = np.random.default_rng(5).random((352, 1216)).astype(np.float32)
    back = kb.backward({"depth_pred": pred, "metadata": e["metadata"]})["depth_pred"]
    assert list(back.shape) == list(gd[f"{tag}.kb.back_shape"]) and float(back.sum()) == float(gd[f"{tag}.kb.back_sum"]) and np.array_equal(back[::37, ::101], gd[f"{tag}.kb.back_probe"])
    ct = build_preprocess({"NAME": "CropTopTo", "IMG_H": 320})
    e = ct.forward(_clone(d))
    assert list(e["img"].shape) == list(gd[f"{tag}.ct.img_shape"]) and np.array_equal(e["intrinsics"], gd[f"{tag}.ct.K"]) and float(e["depth"].sum()) == float(gd[f"{tag}.ct.depth_sum"])
    assert [e["metadata"][k] for k in ("crop_y_start", "h_before_crop", "w_before_crop")] == list(gd[f"{tag}.ct.meta"])
    back = ct.backward({"depth_pred": np.ones((320, W), np.float32), "metadata": e["metadata"]})["depth_pred"]
    assert np.array_equal(back.sum(1).astype(np.float64), gd[f"{tag}.ct.back_rows"])
    random.seed(1234)                                 # the same stream of draws as the generator: RandomCrop (x, then y), then 16 flips
    rc = build_preprocess({"NAME": "RandomCrop", "IMG_H": 352, "IMG_W": 704})
    e = rc.forward(_clone(d))
    assert [e["metadata"][k] for k in ("rand_y_start", "rand_x_start", "h_before_rand_crop", "w_before_rand_crop")] == list(gd[f"{tag}.rc.meta"])
    assert np.array_equal(e["intrinsics"], gd[f"{tag}.rc.K"]) and int(e["img"].astype(np.int64).sum()) == int(gd[f"{tag}.rc.img_sum"]) and float(e["depth"].sum()) == float(gd[f"{tag}.rc.depth_sum"])
    flips = [build_preprocess({"NAME": "RandomFlip"}).forward({})["flip"] for _ in range(16)]
    assert flips == [bool(x) for x in gd[f"{tag}.flips"]]
    # RandomCrop.backward pastes the prediction where the crop was taken (the reference's slice only fits a crop at the origin: documented deviation)
    back = rc.backward({"depth_pred": np.ones((352, 704), np.float32), "metadata": e["metadata"]})["depth_pred"]
    y, x = e["metadata"]["rand_y_start"], e["metadata"]["rand_x_start"]
    assert back.shape == (H, W) and back.sum() == 352 * 704 and back[y:y + 352, x:x + 704].all()
    e = build_preprocess({"NAME": "ClipDepth", "MAX_DEPTH": 80}).forward({"depth": d["depth"].copy(), "ctx_depth": [a.copy() for a in d["ctx_depth"]]})
    assert float(max(e["depth"].max(), max(a.max() for a in e["ctx_depth"]))) == float(gd[f"{tag}.clip.max"]) == 80.0 and float(e["depth"].sum()) == float(gd[f"{tag}.clip.sum"])
    for (h, w) in ((192, 640), (96, 320)):
        r = A.resize_depth(d["depth"], (h, w))
        assert np.array_equal(r[::8, ::16], gd[f"{tag}.resize_depth.{h}"]) and int(np.count_nonzero(r)) == int(gd[f"{tag}.resize_depth.{h}.nnz"])
    assert A.resize_depth(d["depth"], d["depth"].shape) is d["depth"]


@pytest.mark.parametrize("tag,kw", [("ctx", {}), ("noctx", dict(FORWARD_CONTEXT=0, BACKWARD_CONTEXT=0)),
                                    ("nodepth_cam3", dict(DEPTH_TYPE="none", USE_CAMS="image_03", STRIDE=2)),
                                    ("bothcams", dict(USE_CAMS=["image_02", "image_03"], BACKWARD_CONTEXT=0))])
def test_kitti_dataset_index_context_calibration_vs_reference(gd, tree, tag, kw):
    root, (raw, depth, split) = tree
    ds = DATASET_REGISTRY.get("KittiDepthV2")(dataset_cfg(raw, depth, split, **kw), None)
    assert ["/".join(m) for m in ds.metadatas] == list(gd[f"ds.{tag}.metadatas"])
    assert ds.valid_inds == list(gd[f"ds.{tag}.valid_inds"]) and len(ds) == len(gd[f"ds.{tag}.valid_inds"])
    assert [json.dumps(c) for c in ds.context_list] == list(gd[f"ds.{tag}.context"])
    items = [ds[i] for i in range(min(len(ds), 3))]
    if items:
        assert np.array_equal(np.stack([it["intrinsics"] for it in items]), gd[f"ds.{tag}.K"]) and items[0]["intrinsics"].dtype == np.float32
    for it, want in zip(items, gd[f"ds.{tag}.item_meta"]):
        got = {k: (os.path.relpath(v, root) if isinstance(v, str) and v.startswith(root) else [os.path.relpath(x, root) if x.startswith(root) else x for x in v] if isinstance(v, list) else v)
               for k, v in it["metadata"].items()}
        assert got == json.loads(str(want))


def test_batch_collator_vs_reference(gd, tree):
    root, (raw, depth, split) = tree
    ds = DATASET_REGISTRY.get("KittiDepthV2")(dataset_cfg(raw, depth, split), None)
    g = torch.Generator().manual_seed(3)
    exs = []
    for i in range(3):
        exs.append({"img": torch.rand(3, 4, 6, generator=g), "img_orig": torch.rand(3, 4, 6, generator=g), "intrinsics": np.full((3, 3), i, np.float32),
                    "depth": np.full((4, 6), i + 0.5, np.float32), "ctx_img": [np.full((3, 4, 6), 10 * i + j, np.float32) for j in range(2)],
                    "ctx_img_orig": [np.full((3, 4, 6), 100 * i + j, np.float32) for j in range(2)], "ctx_depth": [np.full((4, 6), i + j, np.float32) for j in range(2)],
                    "flip": i == 0, "metadata": {"idx": i}, "depth_orig": np.zeros((2, 2), np.float32)})
    b = ds.batch_collator(exs)
    assert sorted(b.keys()) == list(gd["collate.keys"])
    assert [f"{k}:{type(b[k]).__name__}:{type(b[k][0]).__name__ if isinstance(b[k], list) else ''}" for k in sorted(b.keys())] == list(gd["collate.types"])
    assert list(b["img"].shape) == list(gd["collate.img_shape"]) and np.array_equal(b["depth"].numpy(), gd["collate.depth"]) and np.array_equal(b["intrinsics"].numpy(), gd["collate.K"])
    for k, g_ in (("ctx_img", "collate.ctx_img0"),):
        assert np.array_equal(b[k][0], gd[g_])
    assert np.array_equal(b["ctx_img"][1], gd["collate.ctx_img1"]) and np.array_equal(b["ctx_depth"][1], gd["collate.ctx_depth1"]) and np.array_equal(b["ctx_img_orig"][1], gd["collate.ctx_img_orig1"])
    assert b["flip"] is True and bool(gd["collate.flip"]) is True and len(b["metadata"]) == int(gd["collate.n_meta"])


def test_loading_round_trips_png_bytes(tmp_path):
    """LoadImg / LoadDepth decode what was encoded (PNG is lossless): RGB order, 16-bit depth / 255 (sic, loading.py:L59), KEEP_ORIG copy, WITH_CTX lists."""
    from PIL import Image
    r = np.random.default_rng(0)
    img = r.integers(0, 256, (7, 9, 3), dtype=np.uint8); dm = r.integers(0, 65536, (7, 9)).astype(np.uint16)
    Image.fromarray(img).save(tmp_path / "a.png"); Image.fromarray(dm).save(tmp_path / "d.png")
    np.savez(tmp_path / "v.npz", velodyne_depth=dm.astype(np.float64) / 3)
    md = {"img_dir": str(tmp_path / "a.png"), "ctx_img_dir": [str(tmp_path / "a.png")] * 2, "depth_dir": str(tmp_path / "d.png"), "ctx_depth_dir": [str(tmp_path / "v.npz")]}
    d = build_preprocess({"NAME": "LoadImg", "WITH_CTX": True}).forward({"metadata": md})
    assert np.array_equal(d["img"], img) and len(d["ctx_img"]) == 2 and np.array_equal(d["ctx_img"][1], img)
    d = build_preprocess({"NAME": "LoadDepth", "KEEP_ORIG": True, "WITH_CTX": True}).forward(d)
    assert d["depth"].dtype == np.float32 and np.array_equal(d["depth"], dm.astype(np.float32) / 255) and np.array_equal(d["depth_orig"], d["depth"]) and d["depth_orig"] is not d["depth"]
    assert np.array_equal(d["ctx_depth"][0], (dm.astype(np.float64) / 3).astype(np.float32))
    with pytest.raises(AssertionError):
        build_preprocess({"NAME": "LoadImg"}).forward({"metadata": {"img_dir": str(tmp_path / "missing.png")}})
    with pytest.raises(KeyError):
        PREPROCESS_REGISTRY.get("NoSuchStep")


def test_resize_properties_and_backward():
    """cv2's INTER_LINEAR is restated, not pinned: constants stay constant, identity size is the identity, an exact 2x down-sampling of a
    2x2-block image averages each block (pixel centres fall between the taps with weight 1/2 each way), intrinsics / sparse depth / metadata
    follow augmentation.py:L123-160, backward is the nearest-neighbour rule the evaluators use."""
    r = np.random.default_rng(1)
    img = r.integers(0, 256, (12, 20, 3), dtype=np.uint8)
    assert np.array_equal(A.resize_linear_u8(img, 20, 12), img)
    assert np.array_equal(A.resize_linear_u8(np.full((10, 14, 3), 77, np.uint8), 5, 3), np.full((3, 5, 3), 77, np.uint8))
    blocks = r.integers(0, 256, (6, 10, 3)).astype(np.int64)
    up = np.repeat(np.repeat(blocks, 2, 0), 2, 1).astype(np.uint8)
    assert np.array_equal(A.resize_linear_u8(up, 10, 6), blocks.astype(np.uint8))
    a, b = r.integers(0, 256, (4, 6)).astype(np.int64), r.integers(0, 256, (4, 6)).astype(np.int64)
    pair = np.stack([a, b], 2).reshape(4, 12).astype(np.uint8)               # columns alternate a, b: 2x horizontal down-sampling -> rounded mean
    assert np.array_equal(A.resize_linear_u8(pair, 6, 4).astype(np.int64), (a + b + 1) >> 1)
    d = sample(3, 375, 1242)
    rs = build_preprocess({"NAME": "Resize", "IMG_H": 192, "IMG_W": 640})
    e = rs.forward(copy.deepcopy(d))
    assert e["img"].shape == (192, 640, 3) and e["img"].dtype == np.uint8 and all(a.shape == (192, 640, 3) for a in e["ctx_img"])
    K0, K = d["intrinsics"], e["intrinsics"]
    assert np.allclose(K[0, 0], K0[0, 0] * 640 / 1242) and np.allclose(K[0, 2], K0[0, 2] * 640 / 1242) and np.allclose(K[1, 1], K0[1, 1] * 192 / 375) and np.allclose(K[1, 2], K0[1, 2] * 192 / 375)
    assert np.array_equal(e["depth"], A.resize_depth(d["depth"], (192, 640))) and e["metadata"] == {"h_before_resize": 375, "w_before_resize": 1242}
    pred = r.random((192, 640)).astype(np.float32)
    back = rs.backward({"depth_pred": pred, "metadata": e["metadata"]})["depth_pred"]
    from oracle import evaluation as OE
    assert back.shape == (375, 1242) and np.array_equal(back, pred[OE.nearest_map(192, 375)[:, None], OE.nearest_map(640, 1242)[None, :]])


def test_random_image_aug_is_one_parameter_set_for_all_frames_and_deterministic():
    d = sample(4, 24, 40)
    aug = build_preprocess({"NAME": "RandomImageAug"})
    torch.manual_seed(5); random.seed(5)
    e1 = aug.forward(copy.deepcopy(d))
    aug2 = build_preprocess({"NAME": "RandomImageAug"})        # the constructor draws a parameter set too: build before seeding
    torch.manual_seed(5); random.seed(5)
    e2 = aug2.forward(copy.deepcopy(d))
    assert np.array_equal(e1["img"], e2["img"]) and all(np.array_equal(a, b) for a, b in zip(e1["ctx_img"], e2["ctx_img"]))
    assert np.array_equal(e1["img_orig"], d["img"]) and all(np.array_equal(a, b) for a, b in zip(e1["ctx_img_orig"], d["ctx_img"]))
    assert e1["img"].dtype == np.uint8 and e1["img"].shape == d["img"].shape and not np.array_equal(e1["img"], d["img"])
    # the same jitter on every frame: a context frame equal to the target comes out equal
    same = copy.deepcopy(d); same["ctx_img"] = [same["img"].copy(), same["img"].copy()]
    e3 = aug.forward(same)
    assert np.array_equal(e3["ctx_img"][0], e3["img"]) and np.array_equal(e3["ctx_img"][1], e3["img"])
    assert 0.8 <= aug.b <= 1.2 and 0.8 <= aug.c <= 1.2 and 0.8 <= aug.s <= 1.2 and -0.05 <= aug.h <= 0.05 and sorted(aug.fn_idx.tolist()) == [0, 1, 2, 3]
    off = build_preprocess({"NAME": "RandomImageAug", "JITTER_PROB": 0.0}).forward(copy.deepcopy(d))
    assert np.array_equal(off["img"], d["img"]) and "img_orig" in off
    # hue shift by zero and enhancement factors of one are the identity
    from PIL import Image
    im = Image.fromarray(d["img"])
    assert np.array_equal(np.array(A._adjust_hue(im, 0.0)), np.array(im.convert("HSV").convert("RGB")))
    aug.fn_idx, aug.b, aug.c, aug.s, aug.h = torch.tensor([0, 1, 2]), 1.0, 1.0, 1.0, 0.0
    assert np.array_equal(np.array(aug.augment(im)), d["img"])


def _full_cfg(raw, depth, split, train_steps, test_steps, batch=2, workers=0):
    from simpledepthestimation_amd.config import get_project_cfg
    cfg = get_project_cfg("MonoDepth2")
    cfg.DATASETS.TRAIN = dataset_cfg(raw, depth, split, NAME="KittiDepthV2", DEPTH_TYPE="none", PREPROCESS=train_steps)
    cfg.DATASETS.TEST = dataset_cfg(raw, depth, split, NAME="KittiDepthV2", FORWARD_CONTEXT=0, BACKWARD_CONTEXT=0, PREPROCESS=test_steps)
    cfg.SOLVER.IMS_PER_BATCH = batch; cfg.DATALOADER.NUM_WORKERS = workers
    return cfg


def test_loaders_produce_the_batch_dict_contract(tree):
    """End to end over the synthetic tree with the MonoDepth2 chain of Base.yaml (LoadImg+ctx, Resize, RandomFlip, RandomImageAug, ToTensor /
    LoadImg, LoadDepth+orig, ClipDepth, Resize, ToTensor): the batch dict of SURVEY.md 8b, and it feeds the trainer's static-batch logic."""
    root, (raw, depth, split) = tree
    train = [{"NAME": "LoadImg", "WITH_CTX": True}, {"NAME": "Resize", "IMG_W": 32, "IMG_H": 8}, {"NAME": "RandomFlip"}, {"NAME": "RandomImageAug"}, {"NAME": "ToTensor"}]
    test = [{"NAME": "LoadImg"}, {"NAME": "LoadDepth", "KEEP_ORIG": True}, {"NAME": "ClipDepth", "MAX_DEPTH": 80}, {"NAME": "Resize", "IMG_W": 32, "IMG_H": 8}, {"NAME": "ToTensor"}]
    cfg = _full_cfg(raw, depth, split, train, test)
    tl = build_detection_train_loader(cfg)
    batches = list(tl)
    assert len(batches) == len(tl) == len(tl.dataset) // 2 and len(tl.dataset) == 5          # 5 image_02 frames have both neighbours; drop_last
    b = batches[0]
    assert b["img"].shape == (2, 3, 8, 32) and b["img"].dtype == torch.float32 and 0 <= float(b["img"].min()) and float(b["img"].max()) <= 1
    assert b["img_orig"].shape == (2, 3, 8, 32) and isinstance(b["ctx_img"], list) and len(b["ctx_img"]) == 2 and isinstance(b["ctx_img"][0], np.ndarray)
    assert b["ctx_img"][0].shape == (2, 3, 8, 32) and b["ctx_img_orig"][1].shape == (2, 3, 8, 32) and b["ctx_img"][0].dtype == np.float32
    assert b["intrinsics"].shape == (2, 3, 3) and b["intrinsics"].dtype == torch.float32 and isinstance(b["flip"], bool) and len(b["metadata"]) == 2
    assert abs(float(b["intrinsics"][0, 0, 0]) - 721.5377 * 32 / 40) < 1e-3 or abs(float(b["intrinsics"][0, 0, 0]) - 718.856 * 32 / 40) < 1e-3
    from simpledepthestimation_amd.engine.trainer import HipTrainer
    assert HipTrainer._kind(b["ctx_img"]) == "arrays" and HipTrainer._kind(b["flip"]) == "scalar" and HipTrainer._kind(b["metadata"]) == "opaque"
    te = build_detection_test_loader(cfg)
    tb = list(te)
    assert len(tb) == len(te.dataset) == 13 and tb[0]["img"].shape == (1, 3, 8, 32) and tb[0]["depth"].shape == (1, 1, 8, 32)      # 14 image_02 frames, one without depth
    assert isinstance(tb[0]["depth_orig"], list) and tb[0]["depth_orig"][0].shape == (12, 40) and tb[0]["metadata"][0]["h_before_resize"] == 12
    # predictions travel back through the chain to the original size
    pred = te.dataset.get_prediction({"depth_pred": np.ones((8, 32), np.float32), "metadata": tb[0]["metadata"][0]})["depth_pred"]
    assert pred.shape == (12, 40)
    # the prefetcher hands over the same batches (CPU: a plain move)
    pf = list(DevicePrefetcher(batches, "cpu"))
    assert len(pf) == len(batches) and torch.equal(pf[0]["img"], batches[0]["img"]) and torch.is_tensor(pf[0]["ctx_img"][0]) and pf[0]["flip"] == batches[0]["flip"]
    assert list(InferenceSampler(5)) == [0, 1, 2, 3, 4]
    # SAMPLER_TRAIN "TrainingSampler" (data/build.py:L108-109 of the reference): the loader never ends -- more than one epoch of batches comes out
    cfg.DATALOADER.SAMPLER_TRAIN = "TrainingSampler"
    inf = iter(build_detection_train_loader(cfg))
    many = [next(inf) for _ in range(7)]                # 7 batches of 2 from 5 samples: almost three epochs
    assert all(m["img"].shape == (2, 3, 8, 32) for m in many)
    cfg.DATALOADER.SAMPLER_TRAIN = "NoSuchSampler"
    with pytest.raises(ValueError):
        build_detection_train_loader(cfg)


def test_training_sampler_is_the_strided_stream_of_seeded_permutations():
    """samplers/distributed_sampler.py:L12-52: indices[rank::world] of shuffle(range(n)) + shuffle(range(n)) + ... drawn from ONE generator."""
    import itertools
    from simpledepthestimation_amd.data.build import TrainingSampler
    n, seed = 7, 1234
    g = torch.Generator(); g.manual_seed(seed)
    stream = [int(i) for _ in range(6) for i in torch.randperm(n, generator=g)]
    for world, rank in [(1, 0), (2, 0), (2, 1), (3, 2), (4, 1)]:
        s = TrainingSampler(n, seed=seed)
        s._world, s._rank = world, rank
        got = list(itertools.islice(iter(s), 30 // world))
        assert got == stream[rank::world][:len(got)], (world, rank)
    s = TrainingSampler(4, shuffle=False, seed=0)
    assert list(itertools.islice(iter(s), 10)) == [0, 1, 2, 3, 0, 1, 2, 3, 0, 1]
    with pytest.raises(ValueError):
        TrainingSampler(0)


def test_pose_utils_numpy_and_torch_vs_reference(gd):
    """geometry/pose_utils.py: OXTS packet -> (R, t) (Mercator), homogeneous transforms and their inverses vs the reference's functions."""
    from simpledepthestimation_amd.geometry import pose_utils as PU
    for p, R, t in zip(gd["pu.packets"], gd["pu.oxts_R"], gd["pu.oxts_t"]):
        R2, t2 = PU.pose_from_oxts_packet_np(p, np.cos(p[0] * np.pi / 180.0))
        assert np.array_equal(R2, R) and np.array_equal(t2, t)
    with pytest.raises(ValueError):
        PU.pose_from_oxts_packet_np(gd["pu.packets"][0][:7], 1.0)
    for T, Ti in zip(gd["pu.T"], gd["pu.invert_pose_np"]):
        mine = PU.invert_pose_np(T)
        assert np.allclose(mine, Ti, rtol=0, atol=1e-15) and np.allclose(mine @ T, np.eye(4), atol=1e-12)
        assert np.array_equal(PU.T_from_R_t_np(T[:3, :3].copy(), T[:3, 3].copy()), T)
    Tt = torch.from_numpy(gd["pu.T"].astype(np.float32))
    assert torch.allclose(PU.invert_pose(Tt), torch.from_numpy(gd["pu.invert_pose"]), rtol=0, atol=1e-6)
    a = 0.3
    assert np.allclose(PU.rotx_np(a) @ [0, 1, 0], [0, np.cos(a), np.sin(a)]) and np.allclose(PU.roty_np(a) @ [0, 0, 1], [np.sin(a), 0, np.cos(a)])
    assert np.allclose(PU.rotz_np(a) @ [1, 0, 0], [np.cos(a), np.sin(a), 0])


def test_kitti_dataset_with_pose_vs_reference(gd, tree):
    """WITH_POSE: data['pose_gt'] = imu2cam @ inv(origin) @ pose @ inv(imu2cam) from the OXTS packets and the three calibration files."""
    root, (raw, depth, split) = tree
    ds = DATASET_REGISTRY.get("KittiDepthV2")(dataset_cfg(raw, depth, split, WITH_POSE=True, FORWARD_CONTEXT=0, BACKWARD_CONTEXT=0), None)
    poses = np.stack([ds[i]["pose_gt"] for i in range(len(ds))])
    assert poses.dtype == np.float32 and poses.shape == gd["ds.pose.pose_gt"].shape
    assert np.allclose(poses, gd["ds.pose.pose_gt"], rtol=0, atol=1e-6)
    b = ds.batch_collator([ds[0], ds[1]])
    assert isinstance(b["pose_gt"], torch.Tensor) and tuple(b["pose_gt"].shape) == (2, 4, 4)
    first = [i for i in range(len(ds)) if ds.metadatas[ds.valid_inds[i]][3] == "0000000000"]
    assert first and all(np.allclose(poses[i], np.eye(4), atol=1e-6) for i in first)       # the drive's first frame is the origin
