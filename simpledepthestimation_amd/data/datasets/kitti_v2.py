"""KITTI raw + depth reader (reference: detectron2/data/datasets/kitti_v2.py:L15-221 ``KittiDepthV2``).

On-disk formats: the split file lists ``<date>/<date>_drive_<drive>_sync/<cam>/data/<img_id>.png`` entries (whitespace separated); images
are 8-bit PNGs; ``calib_cam_to_cam.txt`` holds ``key: v0 v1 ...`` lines, ``P_rect_0<cam digit>`` (3x4, row-major) gives the intrinsics K;
depth ground truth is a 16-bit PNG (or a velodyne .npz).  A sample survives only if its files exist, and -- with FORWARD_CONTEXT /
BACKWARD_CONTEXT -- if its neighbours at +-STRIDE frames of the same drive and camera are in the list.
WITH_POSE adds data['pose_gt'], the OXTS odometry pose relative to the drive's first frame (false in every config of the two projects)."""
import logging
import os
from collections import defaultdict

import numpy as np
import torch

from ...geometry import pose_utils as PU
from ..build import DATASET_REGISTRY, DatasetBase

logger = logging.getLogger(__name__)


def read_calib(filepath):
    """``key: floats`` lines -> {key: float32 array}; lines that are not numeric (dates) are skipped (kitti_v2.py:L164-177)."""
    out = {}
    with open(filepath, "r") as f:
        for line in f.readlines():
            key, value = line.split(":", 1)
            try:
                out[key] = np.array([float(x) for x in value.split()], dtype=np.float32)
            except ValueError:
                pass
    return out


def parse_split_entry(entry):
    """'2011_09_26/2011_09_26_drive_0002_sync/image_02/data/0000000069.png' -> (date, drive, cam, img_id)."""
    parts = entry.split("/")
    date = parts[0]
    drive = parts[1].replace(f"{date}_drive_", "").replace("_sync", "")
    return date, drive, parts[2], parts[-1].replace(".png", "")


@DATASET_REGISTRY.register()
class KittiDepthV2(DatasetBase):
    def __init__(self, dataset_cfg, cfg):
        super().__init__(dataset_cfg, cfg)
        self.data_root, self.depth_root, self.split_file = dataset_cfg.DATA_ROOT, dataset_cfg.get("DEPTH_ROOT", ""), dataset_cfg.SPLIT
        self.depth_type = dataset_cfg.get("DEPTH_TYPE", "none")
        self.with_depth = self.depth_type != "none"
        self.use_cams = dataset_cfg.get("USE_CAMS", "image_02")
        self.forward_context = dataset_cfg.get("FORWARD_CONTEXT", 0)
        self.backward_context = dataset_cfg.get("BACKWARD_CONTEXT", 0)
        self.stride = dataset_cfg.get("STRIDE", 0)
        self.with_pose = dataset_cfg.get("WITH_POSE", False)     # OXTS odometry ground truth as data['pose_gt'] (false in every config of the two projects)

        metas, count = [], 0
        for line in open(self.split_file, "r"):
            for entry in line.strip().split():
                date, drive, cam, img_id = parse_split_entry(entry)
                count += 1
                if not os.path.isfile(self._get_img_dir(date, drive, cam, img_id)) \
                        or (self.with_depth and not os.path.isfile(self._get_depth_dir(date, drive, cam, img_id))) or cam not in self.use_cams:
                    continue
                metas.append((date, drive, cam, img_id))
        self.metadatas = sorted(metas)
        logger.info("Loaded %d samples", count)
        logger.info("After existence filtering, %d samples left", len(self.metadatas))

        self.context_list = [[] for _ in self.metadatas]
        self.with_context = self.backward_context != 0 or self.forward_context != 0
        if self.with_context:
            self.valid_inds = []
            for idx, (date, drive, cam, img_id) in enumerate(self.metadatas):
                for offset in range(-self.backward_context * self.stride, self.forward_context * self.stride + 1, self.stride):
                    j = idx + offset
                    if offset != 0 and 0 <= j < len(self.metadatas) and self.metadatas[j][:3] == (date, drive, cam) \
                            and int(self.metadatas[j][3]) == int(img_id) + offset:
                        self.context_list[idx].append(j)
                if len(self.context_list[idx]) == self.backward_context + self.forward_context:
                    self.valid_inds.append(idx)
        else:
            self.valid_inds = list(range(len(self.metadatas)))
        logger.info("After context filtering, %d samples left", len(self.valid_inds))
        if not self.metadatas:
            logger.warning("Empty dataset!")
        self.calib_cache = {}

    def __len__(self):
        return len(self.valid_inds)

    def intrinsics(self, date, cam):
        if date not in self.calib_cache:
            self.calib_cache[date] = read_calib(os.path.join(self.data_root, date, "calib_cam_to_cam.txt"))
        Px = np.eye(4, dtype=np.float32)
        Px[:3, :] = np.array(self.calib_cache[date][f"P_rect_0{cam[-1]}"]).reshape([3, 4])
        return Px[:3, :3]

    def __getitem__(self, idx_):
        idx = self.valid_inds[idx_]
        date, drive, cam, img_id = self.metadatas[idx]
        ctx = [self.metadatas[j] for j in self.context_list[idx]]
        data = {"metadata": {"idx": idx, "date": date, "drive": drive, "cam": cam, "img_id": img_id,
                             "img_dir": self._get_img_dir(date, drive, cam, img_id), "depth_dir": self._get_depth_dir(date, drive, cam, img_id),
                             "lidar_dir": self._get_lidar_dir(date, drive, cam, img_id),
                             "ctx_img_dir": [self._get_img_dir(*m) for m in ctx], "ctx_depth_dir": [self._get_depth_dir(*m) for m in ctx],
                             "ctx_lidar_dir": [self._get_lidar_dir(*m) for m in ctx]},
                "intrinsics": self.intrinsics(date, cam)}
        if self.with_pose:
            data["pose_gt"] = self._get_pose(date, drive, img_id)
        return self.preprocess(data)

    def _get_oxts_dir(self, date, drive, img_id):
        return os.path.join(self.data_root, date, f"{date}_drive_{drive}_sync", "oxts", "data", f"{img_id}.txt")

    def _imu2cam(self, date):
        """Rectified camera-0 frame <- IMU: R_rect_00 @ velo_to_cam @ imu_to_velo (kitti_v2.py:L123-131)."""
        key = (date, "imu2cam")
        if key not in self.calib_cache:
            R0 = np.eye(4, dtype=np.float32)
            R0[:3, :3] = read_calib(os.path.join(self.data_root, date, "calib_cam_to_cam.txt"))["R_rect_00"].reshape([3, 3])
            velo = read_calib(os.path.join(self.data_root, date, "calib_velo_to_cam.txt"))
            imu = read_calib(os.path.join(self.data_root, date, "calib_imu_to_velo.txt"))
            self.calib_cache[key] = R0 @ PU.T_from_R_t_np(velo["R"], velo["T"]) @ PU.T_from_R_t_np(imu["R"], imu["T"])
        return self.calib_cache[key]

    def _get_pose(self, date, drive, img_id):
        """Camera pose relative to the drive's first frame from the OXTS packets (kitti_v2.py:L178-195): the Mercator scale is fixed by the first
        frame's latitude, poses are expressed in the rectified camera frame through imu2cam."""
        origin = np.loadtxt(self._get_oxts_dir(date, drive, "0000000000"), delimiter=" ", skiprows=0)
        scale = np.cos(origin[0] * np.pi / 180.0)
        origin_pose = PU.T_from_R_t_np(*PU.pose_from_oxts_packet_np(origin, scale))
        pose = PU.T_from_R_t_np(*PU.pose_from_oxts_packet_np(np.loadtxt(self._get_oxts_dir(date, drive, img_id), delimiter=" ", skiprows=0), scale))
        imu2cam = self._imu2cam(date)
        return (imu2cam @ np.linalg.inv(origin_pose) @ pose @ np.linalg.inv(imu2cam)).astype(np.float32)

    def _get_img_dir(self, date, drive, cam, img_id):
        return os.path.join(self.data_root, date, f"{date}_drive_{drive}_sync", cam, "data", f"{img_id}.png")

    def _get_depth_dir(self, date, drive, cam, img_id):
        if self.depth_type == "none":
            return ""
        if self.depth_type == "velodyne":
            return os.path.join(self.depth_root, date, f"{date}_drive_{drive}_sync", "proj_depth", "velodyne", cam, f"{img_id}.npz")
        if self.depth_type == "groundtruth":
            return os.path.join(self.depth_root, date, f"{date}_drive_{drive}_sync", "proj_depth", "groundtruth", cam, f"{img_id}.png")
        if self.depth_type == "refined":
            return os.path.join(self.depth_root, f"{date}_drive_{drive}_sync", "proj_depth", "groundtruth", cam, f"{img_id}.png")
        raise NotImplementedError(self.depth_type)

    def _get_lidar_dir(self, date, drive, cam, img_id):
        return os.path.join(self.data_root, date, f"{date}_drive_{drive}_sync", "velodyne_points", "data", f"{img_id}.bin")

    def batch_collator(self, batch_list):
        """list of sample dicts -> the batch dict of SURVEY.md 8b: img / img_orig stacked tensors, intrinsics / depth tensors from numpy,
        ctx_img / ctx_img_orig / ctx_depth as lists (one entry per context) of numpy [B, ...] arrays, ONE flip flag for the batch (the first
        sample's), everything else as a list."""
        merged = defaultdict(list)
        for ex in batch_list:
            for k, v in ex.items():
                merged[k].append(v)
        ret = {}
        for key, value in merged.items():
            if key in ("img", "img_orig"):
                ret[key] = torch.stack(value, 0)
            elif key in ("intrinsics", "pose_gt"):
                ret[key] = torch.from_numpy(np.stack(value, 0))
            elif key == "depth":
                ret[key] = torch.from_numpy(np.stack(value, 0)[:, None, ...])
            elif key in ("ctx_img", "ctx_img_orig"):
                arr = np.stack([np.stack(v, 0) for v in value])
                ret[key] = [arr[:, i] for i in range(arr.shape[1])]
            elif key == "ctx_depth":
                arr = np.stack([np.stack(v, 0)[:, None, ...] for v in value])
                ret[key] = [arr[:, i] for i in range(arr.shape[1])]
            elif key == "flip":
                ret[key] = value[0]
            elif key == "aug_params":
                ret[key] = torch.from_numpy(np.stack(value, 0))
            else:
                ret[key] = value
        if "img_u8" in merged:
            # ON_DEVICE chain (an addition of this package): when every frame of the batch has one source size they are stacked here, in the worker, so
            # that the loader's pin-memory thread pins ONE tensor per entry and data/device_aug.py uploads each with one copy; otherwise per-frame lists
            shapes = {v.shape for v in merged["img_u8"]} | {a.shape for v in merged.get("ctx_img_u8", []) for a in v}
            if len(shapes) == 1:
                ret["img_u8"] = torch.from_numpy(np.stack(merged["img_u8"], 0))
                if "ctx_img_u8" in merged:
                    ret["ctx_img_u8"] = [torch.from_numpy(np.stack([v[i] for v in merged["ctx_img_u8"]], 0)) for i in range(len(merged["ctx_img_u8"][0]))]
        return ret
