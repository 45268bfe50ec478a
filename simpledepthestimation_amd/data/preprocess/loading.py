"""File loading steps (reference: detectron2/data/preprocess/loading.py:L25-79 LoadImg, LoadDepth).

The reference decodes with OpenCV (absent here); Pillow decodes the same PNG bytes: an 8-bit RGB PNG gives the array cv2.imread +
COLOR_BGR2RGB gives, a 16-bit grayscale PNG the array cv2.imread(path, -1) gives (PNG decoding is lossless, so the arrays are equal by the
format's definition; tests write PNGs and read them back)."""
import os

import numpy as np

from .build import PREPROCESS_REGISTRY, Preprocess


def read_rgb(path):
    from PIL import Image
    if not os.path.isfile(path):
        raise AssertionError(f"'{path} does not exist!'")       # the reference asserts on cv2.imread's None
    with Image.open(path) as im:
        return np.asarray(im.convert("RGB"), dtype=np.uint8).copy()


def read_depth_png(path):
    """KITTI depth maps are 16-bit PNGs of depth * 256; the reference divides by 255 (sic, loading.py:L59) -- kept."""
    from PIL import Image
    with Image.open(path) as im:
        arr = np.asarray(im)
    if arr.dtype not in (np.uint16, np.int32, np.uint8):
        arr = arr.astype(np.int32)
    return arr.astype(np.float32) / 255


@PREPROCESS_REGISTRY.register()
class LoadImg(Preprocess):
    def __init__(self, cfg):
        super().__init__(cfg)
        self.load_ctx = cfg.get("WITH_CTX", False)

    def forward(self, data_dict):
        data_dict["img"] = read_rgb(data_dict["metadata"]["img_dir"])
        if self.load_ctx:
            data_dict["ctx_img"] = [read_rgb(p) for p in data_dict["metadata"]["ctx_img_dir"]]
        return data_dict


@PREPROCESS_REGISTRY.register()
class LoadDepth(Preprocess):
    def __init__(self, cfg):
        super().__init__(cfg)
        self.load_ctx = cfg.get("WITH_CTX", False)
        self.keep_orig_for_eval = cfg.get("KEEP_ORIG", False)

    @staticmethod
    def _load(depth_dir):
        ext = os.path.splitext(depth_dir)[-1]
        if ext == ".npz":
            return np.load(depth_dir)["velodyne_depth"].astype(np.float32)
        if ext == ".png":
            return read_depth_png(depth_dir)
        raise NotImplementedError(depth_dir)

    def forward(self, data_dict):
        data_dict["depth"] = self._load(data_dict["metadata"]["depth_dir"])
        if self.keep_orig_for_eval:
            data_dict["depth_orig"] = data_dict["depth"].copy()
        if self.load_ctx:
            data_dict["ctx_depth"] = [self._load(p) for p in data_dict["metadata"]["ctx_depth_dir"]]
        return data_dict
