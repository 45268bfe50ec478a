"""Generates tests/golden/data.npz by RUNNING the reference's own data-path code on seeded inputs (build container only).

detectron2/data/preprocess/augmentation.py and datasets/kitti_v2.py cannot be imported (cv2, torchvision, fvcore, easydict are absent), so -- as in
oracle/gen_golden_eval.py -- the definitions that need none of them are compiled, unmodified, from the files' syntax trees into a namespace
holding numpy / random / torch and stand-ins for the registry decorator and the two base classes:
  augmentation.py: resize_depth, KBCrop, CropTopTo, RandomCrop (forward), RandomFlip, ClipDepth        (L14-24, L27-120, L170-240)
  kitti_v2.py:     KittiDepthV2 (split parsing, existence / context filtering, calibration, sample dict, batch_collator, WITH_POSE)   (L15-221)
  geometry/pose_utils.py: rot{x,y,z}_np, OxtsPacket, pose_from_oxts_packet_np, T_from_R_t_np, invert_pose_np, euler2mat, invert_pose   (L7-127, L140-145)
`np.int` (removed from numpy >= 1.24, used at augmentation.py:L22-23) is provided by the namespace's numpy proxy, as numpy 1.19 (the pinned
version, requirements.txt) defined it: the builtin int.
The KITTI directory tree the reader walks is synthetic and rebuilt identically by tests/test_data.py (make_kitti_tree below).

    python -m oracle.gen_golden_data
"""
import ast
import collections
import json
import logging
import os
import random
import sys
import tempfile
import types

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
REF = "/root/reference/detectron2/data"
OUT = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests", "golden", "data.npz")


class _Reg:
    def register(self):
        return lambda c: c


class _Preprocess:
    def __init__(self, cfg):
        self.cfg = cfg


class _DatasetBase:
    def __init__(self, dataset_cfg, cfg):
        self.preprocesses = []

    def preprocess(self, d):
        return d


class Cfg(dict):
    __getattr__ = dict.__getitem__


def _compile(path, names, ns):
    tree = ast.parse(open(path).read(), path)
    def name(n):
        if isinstance(n, (ast.FunctionDef, ast.ClassDef)):
            return n.name
        if isinstance(n, ast.Assign) and len(n.targets) == 1 and isinstance(n.targets[0], ast.Name):      # module constants (OxtsPacket)
            return n.targets[0].id
        return None
    keep = [n for n in tree.body if name(n) in names]
    assert sorted(name(n) for n in keep) == sorted(names), [name(n) for n in keep]
    exec(compile(ast.Module(body=keep, type_ignores=[]), path, "exec"), ns)
    return ns


def reference_preprocess():
    np_proxy = types.ModuleType("numpy_with_int")
    np_proxy.__dict__.update(np.__dict__)
    np_proxy.int = int
    ns = {"np": np_proxy, "random": random, "torch": torch, "PREPROCESS_REGISTRY": _Reg(), "Preprocess": _Preprocess}
    return _compile(os.path.join(REF, "preprocess", "augmentation.py"), ("resize_depth", "KBCrop", "CropTopTo", "RandomCrop", "RandomFlip", "ClipDepth"), ns)


def reference_pose_utils():
    ns = {"np": np, "torch": torch, "namedtuple": collections.namedtuple}
    return _compile(os.path.join(os.path.dirname(REF), "geometry", "pose_utils.py"),
                    ("rotx_np", "roty_np", "rotz_np", "OxtsPacket", "pose_from_oxts_packet_np", "T_from_R_t_np", "invert_pose_np", "euler2mat", "invert_pose"), ns)


def reference_dataset():
    pu = reference_pose_utils()
    ns = {"np": np, "os": os, "torch": torch, "defaultdict": collections.defaultdict, "logger": logging.getLogger("ref"), "DATASET_REGISTRY": _Reg(),
          "DatasetBase": _DatasetBase, "pose_from_oxts_packet_np": pu["pose_from_oxts_packet_np"], "T_from_R_t_np": pu["T_from_R_t_np"]}
    return _compile(os.path.join(REF, "datasets", "kitti_v2.py"), ("KittiDepthV2",), ns)["KittiDepthV2"]


def sample(seed, H, W, nctx=2, with_mask=False):
    r = np.random.default_rng(seed)
    d = {"img": r.integers(0, 256, (H, W, 3), dtype=np.uint8), "depth": np.where(r.random((H, W)) < 0.2, r.random((H, W)) * 90, 0).astype(np.float32),
         "intrinsics": np.array([[721.5, 0, 609.5], [0, 721.5, 172.8], [0, 0, 1]], np.float32), "metadata": {},
         "ctx_img": [r.integers(0, 256, (H, W, 3), dtype=np.uint8) for _ in range(nctx)],
         "ctx_depth": [np.where(r.random((H, W)) < 0.2, r.random((H, W)) * 90, 0).astype(np.float32) for _ in range(nctx)]}
    if with_mask:
        d["mask"] = (r.random((H, W)) < 0.5).astype(np.float32)
    return d


def make_kitti_tree(root):
    """A miniature KITTI raw + refined-depth tree: two dates, three drives, frames with holes (so context filtering bites), one missing depth
    file, a second camera, and the split file.  Images 12 x 40 RGB, depth 16-bit PNGs.  Deterministic."""
    from PIL import Image
    r, ro = np.random.default_rng(42), np.random.default_rng(43)          # ro: the OXTS packets (their own stream: the images keep their values)
    drives = [("2011_09_26", "0001", [0, 1, 2, 3, 5, 6, 7]), ("2011_09_26", "0002", [10, 11, 12]), ("2011_09_28", "0001", [0, 1, 2, 4])]
    entries = []
    for date, drive, frames in drives:
        os.makedirs(os.path.join(root, "raw", date), exist_ok=True)
        with open(os.path.join(root, "raw", date, "calib_cam_to_cam.txt"), "w") as f:
            f.write("calib_time: 09-Jan-2012 13:57:47\ncorner_dist: 9.950000e-02\n")
            f.write("R_rect_00: 9.999239e-01 9.837760e-03 -7.445048e-03 -9.869795e-03 9.999421e-01 -4.278459e-03 7.402527e-03 4.351614e-03 9.999631e-01\n")
            base = 721.5377 if date.endswith("26") else 718.856
            f.write(f"P_rect_02: {base:e} 0.000000e+00 6.095593e+02 4.485728e+01 0.000000e+00 {base:e} 1.728540e+02 2.163791e-01 0.000000e+00 0.000000e+00 1.000000e+00 2.745884e-03\n")
            f.write(f"P_rect_03: {base:e} 0.000000e+00 6.095593e+02 -3.395242e+02 0.000000e+00 {base:e} 1.728540e+02 2.199936e+00 0.000000e+00 0.000000e+00 1.000000e+00 2.729905e-03\n")
        for name in ("calib_velo_to_cam.txt", "calib_imu_to_velo.txt"):       # read (and, without WITH_POSE, ignored) by the reference's __getitem__
            with open(os.path.join(root, "raw", date, name), "w") as f:
                if name == "calib_velo_to_cam.txt":
                    f.write("calib_time: 15-Mar-2012 11:37:16\nR: 7.533745e-03 -9.999714e-01 -6.166020e-04 1.480249e-02 7.280733e-04 -9.998902e-01 9.998621e-01 "
                            "7.523790e-03 1.480755e-02\nT: -4.069766e-03 -7.631618e-02 -2.717806e-01\n")
                else:
                    f.write("calib_time: 25-May-2012 16:47:16\nR: 9.999976e-01 7.553071e-04 -2.035826e-03 -7.854027e-04 9.998898e-01 -1.482298e-02 2.024406e-03 "
                            "1.482454e-02 9.998881e-01\nT: -8.086759e-01 3.195559e-01 -7.997231e-01\n")
        op = os.path.join(root, "raw", date, f"{date}_drive_{drive}_sync", "oxts", "data")
        os.makedirs(op, exist_ok=True)
        lat, lon, alt, yaw = 49.0 + ro.random() * 0.03, 8.4 + ro.random() * 0.05, 110.0 + ro.random() * 10, ro.uniform(-3.0, 3.0)
        for fr in range(0, max(frames) + 1):         # a packet per frame from the drive's first one (the origin of the odometry poses)
            lat, lon, alt, yaw = lat + ro.normal() * 2e-6, lon + ro.normal() * 3e-6, alt + ro.normal() * 0.02, yaw + ro.normal() * 0.01
            vals = [lat, lon, alt, ro.normal() * 0.03, ro.normal() * 0.03, yaw] + list(ro.normal(size=19)) + [4, 10, 4, 4, 0]
            with open(os.path.join(op, f"{fr:010d}.txt"), "w") as f:
                f.write(" ".join(f"{v:.12f}" if i < 25 else str(v) for i, v in enumerate(vals)) + "\n")
        for cam in ("image_02", "image_03"):
            for fr in frames:
                img_id = f"{fr:010d}"
                p = os.path.join(root, "raw", date, f"{date}_drive_{drive}_sync", cam, "data")
                os.makedirs(p, exist_ok=True)
                Image.fromarray(r.integers(0, 256, (12, 40, 3), dtype=np.uint8)).save(os.path.join(p, img_id + ".png"))
                dp = os.path.join(root, "depth", f"{date}_drive_{drive}_sync", "proj_depth", "groundtruth", cam)
                os.makedirs(dp, exist_ok=True)
                if not (date == "2011_09_26" and drive == "0002" and fr == 11 and cam == "image_02"):       # one depth file is missing
                    dm = np.where(r.random((12, 40)) < 0.3, r.integers(256, 20000, (12, 40)), 0).astype(np.uint16)
                    Image.fromarray(dm).save(os.path.join(dp, img_id + ".png"))
                entries.append(f"{date}/{date}_drive_{drive}_sync/{cam}/data/{img_id}.png")
    entries.append("2011_09_26/2011_09_26_drive_0009_sync/image_02/data/0000000000.png")      # listed, not on disk
    order = list(np.random.default_rng(7).permutation(len(entries)))
    with open(os.path.join(root, "split.txt"), "w") as f:
        for i in range(0, len(order), 2):       # two entries per line: the reader splits lines on whitespace
            f.write(" ".join(entries[j] for j in order[i:i + 2]) + "\n")
    return os.path.join(root, "raw"), os.path.join(root, "depth"), os.path.join(root, "split.txt")


def dataset_cfg(raw, depth, split, **kw):
    c = Cfg(DATA_ROOT=raw, DEPTH_ROOT=depth, SPLIT=split, DEPTH_TYPE="refined", FORWARD_CONTEXT=1, BACKWARD_CONTEXT=1, STRIDE=1, WITH_POSE=False, PREPROCESS=[])
    c.update(kw)
    return c


def main():
    ref = reference_preprocess()
    out = {}
    # ---- preprocess steps on seeded samples
    for tag, (H, W) in {"kitti": (375, 1242), "small": (370, 1226)}.items():
        d = sample(11 if tag == "kitti" else 12, H, W, with_mask=True)
        kb = ref["KBCrop"](Cfg())
        e = kb.forward({k: (v.copy() if isinstance(v, np.ndarray) else ([a.copy() for a in v] if isinstance(v, list) else dict(v))) for k, v in d.items()})
        out[f"{tag}.kb.img_sum"] = np.int64(e["img"].astype(np.int64).sum()); out[f"{tag}.kb.img_corner"] = e["img"][:2, :2].copy()
        out[f"{tag}.kb.depth"] = e["depth"][::16, ::32].copy(); out[f"{tag}.kb.K"] = e["intrinsics"].copy()
        out[f"{tag}.kb.mask_sum"] = np.float64(e["mask"].sum()); out[f"{tag}.kb.ctx_sum"] = np.int64(sum(int(a.astype(np.int64).sum()) for a in e["ctx_img"]))
        out[f"{tag}.kb.meta"] = np.array([e["metadata"][k] for k in ("kb_y_start", "kb_x_start", "h_before_kb_crop", "w_before_kb_crop")])
        pred = np.random.default_rng(5).random((352, 1216)).astype(np.float32)
        back = kb.backward({"depth_pred": pred, "metadata": e["metadata"]})["depth_pred"]
        out[f"{tag}.kb.back_shape"] = np.array(back.shape); out[f"{tag}.kb.back_sum"] = np.float64(back.sum()); out[f"{tag}.kb.back_probe"] = back[::37, ::101].copy()
        ct = ref["CropTopTo"](Cfg(IMG_H=320))
        e = ct.forward({k: (v.copy() if isinstance(v, np.ndarray) else ([a.copy() for a in v] if isinstance(v, list) else dict(v))) for k, v in d.items()})
        out[f"{tag}.ct.img_shape"] = np.array(e["img"].shape); out[f"{tag}.ct.K"] = e["intrinsics"].copy(); out[f"{tag}.ct.depth_sum"] = np.float64(e["depth"].sum())
        out[f"{tag}.ct.meta"] = np.array([e["metadata"][k] for k in ("crop_y_start", "h_before_crop", "w_before_crop")])
        back = ct.backward({"depth_pred": np.ones((320, W), np.float32), "metadata": e["metadata"]})["depth_pred"]
        out[f"{tag}.ct.back_rows"] = back.sum(1).astype(np.float64)
        random.seed(1234)
        rc = ref["RandomCrop"](Cfg(IMG_H=352, IMG_W=704))
        e = rc.forward({k: (v.copy() if isinstance(v, np.ndarray) else ([a.copy() for a in v] if isinstance(v, list) else dict(v))) for k, v in d.items()})
        out[f"{tag}.rc.meta"] = np.array([e["metadata"][k] for k in ("rand_y_start", "rand_x_start", "h_before_rand_crop", "w_before_rand_crop")])
        out[f"{tag}.rc.K"] = e["intrinsics"].copy(); out[f"{tag}.rc.img_sum"] = np.int64(e["img"].astype(np.int64).sum()); out[f"{tag}.rc.depth_sum"] = np.float64(e["depth"].sum())
        out[f"{tag}.flips"] = np.array([ref["RandomFlip"](Cfg()).forward({})["flip"] for _ in range(16)])
        e = ref["ClipDepth"](Cfg(MAX_DEPTH=80)).forward({"depth": d["depth"].copy(), "ctx_depth": [a.copy() for a in d["ctx_depth"]]})
        out[f"{tag}.clip.max"] = np.float64(max(e["depth"].max(), max(a.max() for a in e["ctx_depth"]))); out[f"{tag}.clip.sum"] = np.float64(e["depth"].sum())
        for (h, w) in ((192, 640), (96, 320)):
            out[f"{tag}.resize_depth.{h}"] = ref["resize_depth"](d["depth"], (h, w))[::8, ::16].copy()
            out[f"{tag}.resize_depth.{h}.nnz"] = np.int64(np.count_nonzero(ref["resize_depth"](d["depth"], (h, w))))
    # ---- dataset on the synthetic tree
    Ref = reference_dataset()
    with tempfile.TemporaryDirectory() as tmp:
        raw, depth, split = make_kitti_tree(tmp)
        for tag, kw in {"ctx": {}, "noctx": dict(FORWARD_CONTEXT=0, BACKWARD_CONTEXT=0), "nodepth_cam3": dict(DEPTH_TYPE="none", USE_CAMS="image_03", STRIDE=2),
                        "bothcams": dict(USE_CAMS=["image_02", "image_03"], BACKWARD_CONTEXT=0)}.items():
            ds = Ref(dataset_cfg(raw, depth, split, **kw), None)
            out[f"ds.{tag}.metadatas"] = np.array(["/".join(m) for m in ds.metadatas])
            out[f"ds.{tag}.valid_inds"] = np.array(ds.valid_inds, np.int64)
            out[f"ds.{tag}.context"] = np.array([json.dumps(c) for c in ds.context_list])
            items = [ds[i] for i in range(min(len(ds), 3))]
            out[f"ds.{tag}.K"] = np.stack([it["intrinsics"] for it in items]) if items else np.zeros((0, 3, 3), np.float32)
            out[f"ds.{tag}.item_meta"] = np.array([json.dumps({k: (v if not isinstance(v, str) else os.path.relpath(v, tmp) if v.startswith(tmp) else v) if not isinstance(v, list)
                                                               else [os.path.relpath(x, tmp) if x.startswith(tmp) else x for x in v]
                                                               for k, v in it["metadata"].items()}, sort_keys=True) for it in items])
        ds = Ref(dataset_cfg(raw, depth, split, WITH_POSE=True, FORWARD_CONTEXT=0, BACKWARD_CONTEXT=0), None)
        out["ds.pose.pose_gt"] = np.stack([ds[i]["pose_gt"] for i in range(len(ds))])
        # batch_collator on synthetic sample dicts of the shapes the chain produces
        ds = Ref(dataset_cfg(raw, depth, split), None)
        g = torch.Generator().manual_seed(3)
        exs = []
        for i in range(3):
            exs.append({"img": torch.rand(3, 4, 6, generator=g), "img_orig": torch.rand(3, 4, 6, generator=g), "intrinsics": np.full((3, 3), i, np.float32),
                        "depth": np.full((4, 6), i + 0.5, np.float32), "ctx_img": [np.full((3, 4, 6), 10 * i + j, np.float32) for j in range(2)],
                        "ctx_img_orig": [np.full((3, 4, 6), 100 * i + j, np.float32) for j in range(2)], "ctx_depth": [np.full((4, 6), i + j, np.float32) for j in range(2)],
                        "flip": i == 0, "metadata": {"idx": i}, "depth_orig": np.zeros((2, 2), np.float32)})
        b = ds.batch_collator(exs)
        out["collate.keys"] = np.array(sorted(b.keys()))
        out["collate.types"] = np.array([f"{k}:{type(b[k]).__name__}:{type(b[k][0]).__name__ if isinstance(b[k], list) else ''}" for k in sorted(b.keys())])
        out["collate.img_shape"] = np.array(b["img"].shape); out["collate.depth"] = b["depth"].numpy(); out["collate.K"] = b["intrinsics"].numpy()
        out["collate.ctx_img0"] = b["ctx_img"][0]; out["collate.ctx_img1"] = b["ctx_img"][1]; out["collate.ctx_depth1"] = b["ctx_depth"][1]
        out["collate.ctx_img_orig1"] = b["ctx_img_orig"][1]; out["collate.flip"] = np.array(b["flip"]); out["collate.n_meta"] = np.int64(len(b["metadata"]))
    # ---- geometry/pose_utils.py on seeded values
    pu = reference_pose_utils()
    r = np.random.default_rng(21)
    out["pu.angles"] = r.uniform(-3.0, 3.0, (6, 3)).astype(np.float32)
    out["pu.euler2mat"] = pu["euler2mat"](torch.from_numpy(out["pu.angles"])).numpy()
    T = np.stack([pu["T_from_R_t_np"](pu["rotz_np"](a[2]).dot(pu["roty_np"](a[1]).dot(pu["rotx_np"](a[0]))), r.normal(size=3) * 5) for a in out["pu.angles"].astype(np.float64)])
    out["pu.T"] = T
    out["pu.invert_pose_np"] = np.stack([pu["invert_pose_np"](t) for t in T])
    out["pu.invert_pose"] = pu["invert_pose"](torch.from_numpy(T.astype(np.float32))).numpy()
    packets = np.concatenate([np.stack([r.uniform(-80, 80, 4), r.uniform(-180, 180, 4), r.uniform(0, 500, 4)], 1), r.uniform(-3, 3, (4, 3)), r.normal(size=(4, 24))], 1)
    out["pu.packets"] = packets
    Rt = [pu["pose_from_oxts_packet_np"](p, np.cos(p[0] * np.pi / 180.0)) for p in packets]
    out["pu.oxts_R"] = np.stack([x[0] for x in Rt]); out["pu.oxts_t"] = np.stack([x[1] for x in Rt])
    np.savez_compressed(OUT, **out)
    print("wrote", OUT, os.path.getsize(OUT), "bytes,", len(out), "arrays")


if __name__ == "__main__":
    main()
