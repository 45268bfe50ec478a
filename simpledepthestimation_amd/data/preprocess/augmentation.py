"""Geometric / photometric preprocess steps (reference: detectron2/data/preprocess/augmentation.py:L27-266): KBCrop, CropTopTo, Resize,
RandomCrop, RandomFlip, ClipDepth, RandomImageAug, with the backward() of the steps the evaluators undo.

Pinned by goldens produced from the reference's own class bodies (tests/golden/data.npz, oracle/gen_golden_data.py): every step whose
arithmetic is numpy / random only.  NOT pinned (the libraries are absent here and in the reference checkout, "parity unpinned"):
  * Resize of images: cv2.resize(INTER_LINEAR) on uint8 -- restated below from OpenCV's published fixed-point algorithm (11-bit
    coefficients, two passes, rounding (x + 2) >> 2 after two >> 4 / >> 16 shifts); masks / predictions use OpenCV's nearest rule;
  * RandomImageAug: torchvision's ColorJitter functions on PIL images -- restated with the same Pillow calls (ImageEnhance.Brightness /
    Contrast / Color, HSV hue shift in uint8), parameters drawn by the same torch calls in the same order."""
import random

import numpy as np
import torch

from .build import PREPROCESS_REGISTRY, Preprocess

_CROPPED = ("depth", "mask")
_CROPPED_LISTS = ("ctx_img", "ctx_depth", "ctx_mask")


def resize_depth(depth, dst_size):
    """Sparse depth: scatter the valid pixels to their scaled positions (augmentation.py:L14-24)."""
    if depth.shape[-2] == dst_size[-2] and depth.shape[-1] == dst_size[-1]:
        return depth
    H, W = depth.shape
    y, x = np.nonzero(depth)
    out = np.zeros(dst_size, dtype=np.float32)
    out[(dst_size[0] * y / H).astype(int), (dst_size[1] * x / W).astype(int)] = depth[y, x]
    return out


def _crop(data_dict, ys, xs):
    data_dict["img"] = data_dict["img"][ys, xs]
    for k in _CROPPED:
        if k in data_dict:
            data_dict[k] = data_dict[k][ys, xs]
    for k in _CROPPED_LISTS:
        if k in data_dict:
            data_dict[k] = [a[ys, xs] for a in data_dict[k]]


@PREPROCESS_REGISTRY.register()
class KBCrop(Preprocess):
    """The 352 x 1216 bottom-centre crop of the KITTI benchmark (augmentation.py:L27-74)."""

    def forward(self, data_dict):
        img_h, img_w = data_dict["img"].shape[:2]
        x_start, y_start = int((img_w - 1216) / 2), int(img_h - 352)
        _crop(data_dict, slice(y_start, y_start + 352), slice(x_start, x_start + 1216))
        if "intrinsics" in data_dict:
            data_dict["intrinsics"][0, 2] -= x_start
            data_dict["intrinsics"][1, 2] -= y_start
        data_dict["metadata"].update(kb_y_start=y_start, kb_x_start=x_start, h_before_kb_crop=img_h, w_before_kb_crop=img_w)
        return data_dict

    def backward(self, data_dict):
        pred, md = data_dict["depth_pred"], data_dict["metadata"]
        full = np.zeros((md["h_before_kb_crop"], md["w_before_kb_crop"]), dtype=np.float32)
        full[md["kb_y_start"]:md["kb_y_start"] + pred.shape[-2], md["kb_x_start"]:md["kb_x_start"] + pred.shape[-1]] = pred
        data_dict["depth_pred"] = full
        return data_dict


@PREPROCESS_REGISTRY.register()
class CropTopTo(Preprocess):
    def __init__(self, cfg):
        super().__init__(cfg)
        self.height = cfg.IMG_H

    def forward(self, data_dict):
        img_h, img_w = data_dict["img"].shape[:2]
        y_start = int(img_h - self.height)
        _crop(data_dict, slice(y_start, None), slice(None))
        if "intrinsics" in data_dict:
            data_dict["intrinsics"][1, 2] -= y_start
        data_dict["metadata"].update(crop_y_start=y_start, h_before_crop=img_h, w_before_crop=img_w)
        return data_dict

    def backward(self, data_dict):
        pred, md = data_dict["depth_pred"], data_dict["metadata"]
        full = np.zeros((md["h_before_crop"], md["w_before_crop"]), dtype=np.float32)
        full[md["crop_y_start"]:] = pred
        data_dict["depth_pred"] = full
        return data_dict


def _linear_coefs(src, dst):
    """OpenCV INTER_LINEAR source taps and 11-bit fixed-point weights for one axis (resize.cpp: resizeGeneric_ coefficient setup)."""
    scale = src / dst
    f = (np.arange(dst, dtype=np.float64) + 0.5) * scale - 0.5
    s = np.floor(f).astype(np.int64)
    f = (f - s).astype(np.float32)
    lo = s < 0
    f[lo], s[lo] = 0.0, 0
    hi = s >= src - 1
    f[hi], s[hi] = 0.0, src - 1
    w1 = np.rint(f.astype(np.float64) * 2048).astype(np.int64)          # cvRound(f * INTER_RESIZE_COEF_SCALE), saturate_cast<short>
    w0 = np.rint((1.0 - f.astype(np.float64)) * 2048).astype(np.int64)
    return s, np.minimum(s + 1, src - 1), w0, w1


def resize_linear_u8(img, dst_w, dst_h):
    """cv2.resize(img, (dst_w, dst_h), interpolation=cv2.INTER_LINEAR) for uint8 HWC images, restated from the published algorithm
    (horizontal pass into int32 with 11-bit weights, vertical pass ((b0 * (S0 >> 4)) >> 16) + ((b1 * (S1 >> 4)) >> 16) + 2) >> 2)."""
    H, W = img.shape[:2]
    if (H, W) == (dst_h, dst_w):
        return img.copy()
    x0, x1, a0, a1 = _linear_coefs(W, dst_w)
    y0, y1, b0, b1 = _linear_coefs(H, dst_h)
    src = img.astype(np.int64)
    if src.ndim == 2:
        src = src[:, :, None]
    rows = src[:, x0, :] * a0[None, :, None] + src[:, x1, :] * a1[None, :, None]          # [H, dst_w, C]
    out = (((b0[:, None, None] * (rows[y0] >> 4)) >> 16) + ((b1[:, None, None] * (rows[y1] >> 4)) >> 16) + 2) >> 2
    out = np.clip(out, 0, 255).astype(np.uint8)
    return out[:, :, 0] if img.ndim == 2 else out


def resize_nearest(arr, dst_w, dst_h):
    """cv2.resize(..., INTER_NEAREST): source index = min(floor(dst * src / dst_size), src - 1)."""
    H, W = arr.shape[:2]
    ys = np.minimum(np.floor(np.arange(dst_h) * (H / dst_h)).astype(np.int64), H - 1)
    xs = np.minimum(np.floor(np.arange(dst_w) * (W / dst_w)).astype(np.int64), W - 1)
    return arr[ys[:, None], xs[None, :]]


@PREPROCESS_REGISTRY.register()
class Resize(Preprocess):
    """ON_DEVICE: True (an addition of this package) leaves the uint8 frames at their source size -- the device pipeline (data/device_aug.py,
    sde_image_prep_u8) resizes them with the same fixed-point rule after the host-to-device copy -- and still rescales the intrinsics, the sparse
    depth and the masks here (they are small and stay on the host path)."""

    def __init__(self, cfg):
        super().__init__(cfg)
        self.img_h, self.img_w = cfg.IMG_H, cfg.IMG_W
        self.on_device = bool(cfg.get("ON_DEVICE", False))

    def forward(self, data_dict):
        H, W, _ = data_dict["img"].shape
        if self.on_device:
            data_dict["device_resize"] = (self.img_h, self.img_w)
            return self._forward_rest(data_dict, H, W, images=False)
        data_dict["img"] = resize_linear_u8(data_dict["img"], self.img_w, self.img_h)
        return self._forward_rest(data_dict, H, W, images=True)

    def _forward_rest(self, data_dict, H, W, images):
        if "intrinsics" in data_dict:
            K = data_dict["intrinsics"]
            K[0, 0] *= self.img_w / W; K[0, 2] *= self.img_w / W
            K[1, 1] *= self.img_h / H; K[1, 2] *= self.img_h / H
        if "depth" in data_dict:
            data_dict["depth"] = resize_depth(data_dict["depth"], (self.img_h, self.img_w))
        if "mask" in data_dict:
            data_dict["mask"] = resize_nearest(data_dict["mask"], self.img_w, self.img_h)
        if images and "ctx_img" in data_dict:
            data_dict["ctx_img"] = [resize_linear_u8(a, self.img_w, self.img_h) for a in data_dict["ctx_img"]]
        if "ctx_depth" in data_dict:
            data_dict["ctx_depth"] = [resize_depth(d, (self.img_h, self.img_w)) for d in data_dict["ctx_depth"]]
        if "ctx_mask" in data_dict:
            data_dict["ctx_mask"] = [resize_nearest(m, self.img_w, self.img_h) for m in data_dict["ctx_mask"]]
        data_dict["metadata"].update(h_before_resize=H, w_before_resize=W)
        return data_dict

    def backward(self, data_dict):
        md = data_dict["metadata"]
        data_dict["depth_pred"] = resize_nearest(data_dict["depth_pred"], md["w_before_resize"], md["h_before_resize"])
        return data_dict


@PREPROCESS_REGISTRY.register()
class RandomCrop(Preprocess):
    def __init__(self, cfg):
        super().__init__(cfg)
        self.img_h, self.img_w = cfg.IMG_H, cfg.IMG_W

    def forward(self, data_dict):
        img_h, img_w = data_dict["img"].shape[:2]
        assert img_h >= self.img_h and img_w >= self.img_w
        x_start = random.randint(0, img_w - self.img_w)          # x first, then y: the draw order of augmentation.py:L181-182
        y_start = random.randint(0, img_h - self.img_h)
        _crop(data_dict, slice(y_start, y_start + self.img_h), slice(x_start, x_start + self.img_w))
        if "intrinsics" in data_dict:
            data_dict["intrinsics"][0, 2] -= x_start
            data_dict["intrinsics"][1, 2] -= y_start
        data_dict["metadata"].update(rand_y_start=y_start, rand_x_start=x_start, h_before_rand_crop=img_h, w_before_rand_crop=img_w)
        return data_dict

    def backward(self, data_dict):
        pred, md = data_dict["depth_pred"], data_dict["metadata"]
        x, y = md["rand_x_start"], md["rand_y_start"]
        full = np.zeros((md["h_before_rand_crop"], md["w_before_rand_crop"]), dtype=np.float32)
        full[y:y + pred.shape[-2], x:x + pred.shape[-1]] = pred      # the reference slices [y:h_pred, x:w_pred] (L215), which only fits a crop at the origin
        data_dict["depth_pred"] = full
        return data_dict


@PREPROCESS_REGISTRY.register()
class RandomFlip(Preprocess):
    """Only draws the flag: the flip itself happens inside the depth net on the device (DepthResNet.py:L52-60)."""

    def forward(self, data_dict):
        data_dict["flip"] = random.random() > 0.5
        return data_dict


@PREPROCESS_REGISTRY.register()
class ClipDepth(Preprocess):
    def __init__(self, cfg):
        super().__init__(cfg)
        self.max_depth = cfg.MAX_DEPTH

    def forward(self, data_dict):
        if "depth" in data_dict:
            data_dict["depth"] = np.clip(data_dict["depth"], 0, self.max_depth)
        if "ctx_depth" in data_dict:
            data_dict["ctx_depth"] = [np.clip(d, 0, self.max_depth) for d in data_dict["ctx_depth"]]
        return data_dict


def _adjust_hue(img, hue_factor):
    """torchvision.transforms.functional_pil.adjust_hue: H channel of the HSV image shifted by uint8(hue_factor * 255), wrapping."""
    from PIL import Image
    if not -0.5 <= hue_factor <= 0.5:
        raise ValueError(f"hue_factor ({hue_factor}) is not in [-0.5, 0.5].")
    if img.mode in {"L", "1", "I", "F"}:
        return img
    h, s, v = img.convert("HSV").split()
    np_h = np.array(h, dtype=np.uint8)
    shift = int(hue_factor * 255) % 256      # np.uint8(hue_factor * 255) as the numpy the reference pins computes it: truncate towards zero, wrap
    with np.errstate(over="ignore"):
        np_h += np.uint8(shift)
    return Image.merge("HSV", (Image.fromarray(np_h, "L"), s, v)).convert(img.mode)


@PREPROCESS_REGISTRY.register()
class RandomImageAug(Preprocess):
    """Colour jitter (brightness, contrast, saturation, hue in a random order) with ONE parameter set for the target and its context frames;
    keeps the un-jittered frames as img_orig / ctx_img_orig for the photometric loss (augmentation.py:L229-266)."""

    def __init__(self, cfg):
        super().__init__(cfg)
        self.jitter_prob = cfg.get("JITTER_PROB", 1.0)
        p = cfg.get("JITTER_PARAMS", (0.2, 0.2, 0.2, 0.05))
        self.brightness = [max(1 - float(p[0]), 0.0), 1 + float(p[0])]
        self.contrast = [max(1 - float(p[1]), 0.0), 1 + float(p[1])]
        self.saturation = [max(1 - float(p[2]), 0.0), 1 + float(p[2])]
        self.hue = [-float(p[3]), float(p[3])]
        self.on_device = bool(cfg.get("ON_DEVICE", False))
        self.fn_idx = self.b = self.c = self.s = self.h = None
        self.get_params()

    def get_params(self):
        self.fn_idx = torch.randperm(4)
        self.b = float(torch.empty(1).uniform_(self.brightness[0], self.brightness[1]))
        self.c = float(torch.empty(1).uniform_(self.contrast[0], self.contrast[1]))
        self.s = float(torch.empty(1).uniform_(self.saturation[0], self.saturation[1]))
        self.h = float(torch.empty(1).uniform_(self.hue[0], self.hue[1]))

    def augment(self, img):
        from PIL import ImageEnhance
        for fn_id in self.fn_idx:
            fn_id = int(fn_id)
            if fn_id == 0:
                img = ImageEnhance.Brightness(img).enhance(self.b)
            elif fn_id == 1:
                img = ImageEnhance.Contrast(img).enhance(self.c)
            elif fn_id == 2:
                img = ImageEnhance.Color(img).enhance(self.s)
            elif fn_id == 3:
                img = _adjust_hue(img, self.h)
        return img

    def forward(self, data_dict):
        if self.on_device:
            # ON_DEVICE: True (an addition of this package): draw the SAME random numbers in the same order and hand the raw uint8 frames plus the
            # parameters to the device pipeline (data/device_aug.py), which produces img / img_orig / ctx_img / ctx_img_orig after the copy to HBM
            params = np.array([1.0, 1.0, 1.0, 0.0, -1.0, -1.0, -1.0, -1.0], dtype=np.float32)
            if random.random() < self.jitter_prob:
                self.get_params()
                params = np.array([self.b, self.c, self.s, self.h] + [float(i) for i in self.fn_idx], dtype=np.float32)
            data_dict["img_u8"] = np.ascontiguousarray(data_dict.pop("img"))
            if "ctx_img" in data_dict:
                data_dict["ctx_img_u8"] = [np.ascontiguousarray(a) for a in data_dict.pop("ctx_img")]
            data_dict["aug_params"] = params
            data_dict.setdefault("device_resize", tuple(data_dict["img_u8"].shape[:2]))
            return data_dict
        from PIL import Image
        data_dict["img_orig"] = data_dict["img"].copy()
        if "ctx_img" in data_dict:
            data_dict["ctx_img_orig"] = [a.copy() for a in data_dict["ctx_img"]]
        if random.random() < self.jitter_prob:
            self.get_params()
            data_dict["img"] = np.array(self.augment(Image.fromarray(data_dict["img"])))
            if "ctx_img" in data_dict:
                data_dict["ctx_img"] = [np.array(self.augment(Image.fromarray(a))) for a in data_dict["ctx_img"]]
        return data_dict
