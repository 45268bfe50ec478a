"""Device-side image pipeline: uint8 frames -> resized, colour-jittered fp32 batch entries on the GPU (sde_image_prep_u8, csrc/imageprep.hip).

The CPU chain of the reference (detectron2/data/preprocess/augmentation.py:L124-166 `Resize`, L229-266 `RandomImageAug`, formating.py `ToTensor`)
turns every 375 x 1242 camera frame into fp32 [3, 192, 640] tensors on the host: 3 frames x 12 samples per step and GPU for MonoDepth2, i.e. the
data loader's workers resize and jitter ~100 frames per 5 ms step of one MI355X, and the batch crosses PCIe as fp32 (106 MB per step).  With
``ON_DEVICE: True`` on the `Resize` and `RandomImageAug` steps the workers only decode, draw the random parameters (same RNG calls in the same
order) and rescale the small entries; the frames cross PCIe as uint8 at source size (50 MB per step) and this module produces `img`, `img_orig`,
`ctx_img`, `ctx_img_orig` on the device, bit-identical to the CPU chain (tests/test_device_aug.py).
"""
import numpy as np
import torch

from ..hip import lib as L
from .preprocess.augmentation import _linear_coefs

_P, _I = L.c_void_p, L.c_int
L.register_protos({"sde_image_prep_u8": ([_P, _I, _I, _I, _I, _I, _P, _P, _P, _P, _P, _P, _P], L.c_int),
                   "sde_image_prep_u8_multi": ([_P, _I, _I, _I, _I, _I, _I, _P, _P, _P, _P, _P, _P, _P], L.c_int)})
PREP_MAX_GROUPS = 4


def tap_table(src, dst):
    """[dst][4] int32 = (tap 0, tap 1, weight 0, weight 1) of OpenCV's fixed-point INTER_LINEAR along one axis (resize_linear_u8's coefficients)."""
    s0, s1, w0, w1 = _linear_coefs(src, dst)
    return torch.from_numpy(np.stack([s0, s1, w0, w1], 1).astype(np.int32))


class DeviceImageAug:
    """batch dict with the raw entries of the ON_DEVICE chain (`img_u8`: list of B uint8 [Hs,Ws,3] frames, `ctx_img_u8`: list of B lists of context
    frames, `aug_params`: list of B float32 [8], `device_resize`: (h, w)) -> the same dict with `img`, `img_orig` ([B,3,h,w] fp32 device tensors),
    `ctx_img`, `ctx_img_orig` (lists of those) in their place.  Frames are grouped by source size (KITTI drives differ by a few pixels): one
    pair of launches per size.  Everything runs on the current stream; nothing synchronises."""

    def __init__(self, device):
        self.device = torch.device(device)
        self._taps = {}
        self._own = {}

    def taps(self, src, dst):
        key = (src, dst)
        if key not in self._taps:
            self._taps[key] = tap_table(src, dst).to(self.device)
        return self._taps[key]

    @staticmethod
    def _buf(bufs, key, shape, dtype, device):
        """A persistent tensor of the prefetcher's slot (bufs: dict) -- or a fresh one when called without a slot."""
        if bufs is None:
            return torch.empty(shape, dtype=dtype, device=device)
        b = bufs.get(key)
        if b is None or tuple(b.shape) != tuple(shape) or b.dtype != dtype:
            b = bufs[key] = torch.empty(shape, dtype=dtype, device=device)
        return b

    def prep(self, frames, params, h, w, bufs=None, tag="x", out=None):
        """frames: uint8 device tensor [N,Hs,Ws,3]; params: float32 device tensor [N,8] -> (img [N,3,h,w], orig [N,3,h,w]).
        out: optional (img, orig) destination tensors (the captured step's static inputs: the kernels then write them in place)."""
        if frames.dtype != torch.uint8 or frames.dim() != 4 or frames.shape[3] != 3 or not frames.is_cuda:
            raise L.SdeHipError("DeviceImageAug: frames must be a uint8 [N,H,W,3] device tensor")
        N, Hs, Ws, _ = frames.shape
        frames, params = frames.contiguous(), params.contiguous().float()
        ok = lambda t: torch.is_tensor(t) and t.is_cuda and t.dtype == torch.float32 and tuple(t.shape) == (N, 3, h, w) and t.is_contiguous()
        if out is not None and ok(out[0]) and ok(out[1]):
            img, orig = out
        else:
            img = self._buf(bufs, ("aug.img", tag), (N, 3, h, w), torch.float32, self.device)
            orig = self._buf(bufs, ("aug.orig", tag), (N, 3, h, w), torch.float32, self.device)
        lsum = self._buf(bufs, ("aug.lsum", tag), (N,), torch.int32, self.device)
        L.check(L.lib().sde_image_prep_u8(L.ptr(frames), N, Hs, Ws, h, w, L.ptr(self.taps(Ws, w)), L.ptr(self.taps(Hs, h)), L.ptr(params), L.ptr(lsum),
                                          L.ptr(img), L.ptr(orig), L.stream()), "sde_image_prep_u8")
        return img, orig

    def prep_multi(self, frames_list, params, h, w, bufs=None, tag="x", outs=None):
        """G <= 4 uint8 device tensors [N,Hs,Ws,3] of ONE shape sharing the per-sample parameters (target + context frames of a batch) in one pair of
        launches -> [(img, orig)] per tensor.  outs: optional list of (img, orig) destinations (or None entries)."""
        G = len(frames_list)
        f0 = frames_list[0]
        if not (1 <= G <= PREP_MAX_GROUPS) or any(f.dtype != torch.uint8 or f.dim() != 4 or f.shape != f0.shape or not f.is_cuda for f in frames_list):
            raise L.SdeHipError("DeviceImageAug: prep_multi takes 1-4 uint8 [N,H,W,3] device tensors of one shape")
        N, Hs, Ws, _ = f0.shape
        frames_list = [f.contiguous() for f in frames_list]
        params = params.contiguous().float()
        ok = lambda t: torch.is_tensor(t) and t.is_cuda and t.dtype == torch.float32 and tuple(t.shape) == (N, 3, h, w) and t.is_contiguous()
        res = []
        for g in range(G):
            o = outs[g] if outs is not None else None
            if o is not None and ok(o[0]) and ok(o[1]):
                res.append((o[0], o[1]))
            else:
                res.append((self._buf(bufs, ("aug.img", tag, g), (N, 3, h, w), torch.float32, self.device),
                            self._buf(bufs, ("aug.orig", tag, g), (N, 3, h, w), torch.float32, self.device)))
        lsum = self._buf(bufs, ("aug.lsum", tag), (G * N,), torch.int32, self.device)
        arr = lambda ts: (L.c_void_p * G)(*[t.data_ptr() for t in ts])
        L.check(L.lib().sde_image_prep_u8_multi(arr(frames_list), G, N, Hs, Ws, h, w, L.ptr(self.taps(Ws, w)), L.ptr(self.taps(Hs, h)), L.ptr(params), L.ptr(lsum),
                                                arr([r[0] for r in res]), arr([r[1] for r in res]), L.stream()), "sde_image_prep_u8_multi")
        return res

    def _up(self, t, bufs, key):
        """Host tensor -> (the slot's persistent) device tensor, asynchronously from pinned memory."""
        if isinstance(t, np.ndarray):
            t = torch.from_numpy(np.ascontiguousarray(t))
        if t.device.type != "cpu":
            return t
        if not t.is_pinned():
            t = t.pin_memory()
        b = self._buf(bufs, ("aug.up", key), t.shape, t.dtype, self.device)
        b.copy_(t, non_blocking=True)
        return b

    def __call__(self, batch, bufs=None, static=None):
        """static: optional dict holding destination tensors under the output keys (HipTrainer's static batch): the kernels write into them."""
        if "img_u8" not in batch:
            return batch
        if bufs is None:
            bufs = self._own            # persistent scratch when no prefetcher slot is given (consumer-side use: one batch at a time)
        st = static or {}
        dst = lambda k, k2, i=None: ((st[k] if i is None else st[k][i]), (st[k2] if i is None else st[k2][i])) if (k in st and k2 in st) else None
        out = {k: v for k, v in batch.items() if k not in ("img_u8", "ctx_img_u8", "aug_params", "device_resize")}
        size = batch["device_resize"]
        h, w = (size[0] if isinstance(size, (list, tuple)) and isinstance(size[0], (list, tuple)) else size)
        frames = batch["img_u8"]
        params = batch["aug_params"]
        if not torch.is_tensor(params):
            params = torch.as_tensor(np.stack([np.asarray(p, dtype=np.float32) for p in params], 0))
        pd = self._up(params, bufs, "params")
        ctx = batch.get("ctx_img_u8")
        if torch.is_tensor(frames):
            # collated form (every frame of the batch has one source size; the collator stacked them, the loader pinned them): target frames
            # [B,Hs,Ws,3], contexts as a list of such tensors -- one upload and one pair of launches per tensor, outputs are the batch entries
            dev_frames = [self._up(frames, bufs, "img")] + [self._up(c, bufs, ("ctx", i)) for i, c in enumerate(ctx or [])]
            dsts = [dst("img", "img_orig")] + [dst("ctx_img", "ctx_img_orig", i) for i in range(len(dev_frames) - 1)]
            if len(dev_frames) <= PREP_MAX_GROUPS and all(f.shape == dev_frames[0].shape for f in dev_frames):
                res = self.prep_multi(dev_frames, pd, h, w, bufs, "batch", dsts)            # target + context frames: ONE pair of launches
            else:
                res = [self.prep(f, pd, h, w, bufs, ("t", i), dsts[i]) for i, f in enumerate(dev_frames)]
            out["img"], out["img_orig"] = res[0]
            if ctx is not None:
                out["ctx_img"], out["ctx_img_orig"] = [r[0] for r in res[1:]], [r[1] for r in res[1:]]
            return out
        B = len(frames)
        nctx = len(ctx[0]) if ctx is not None else 0
        # per-frame lists (source sizes differ inside the batch): slot s of sample b: s = 0 target, 1.. contexts; one pair of launches per size
        slots = [[frames[b]] + (list(ctx[b]) if nctx else []) for b in range(B)]
        img = torch.empty(B, 1 + nctx, 3, h, w, device=self.device)
        orig = torch.empty(B, 1 + nctx, 3, h, w, device=self.device)
        groups = {}
        for b in range(B):
            for s_, f in enumerate(slots[b]):
                groups.setdefault(tuple(f.shape[:2]), []).append((b, s_, f))
        for (Hs, Ws), items in groups.items():
            fr = self._up(np.stack([np.asarray(f) if not torch.is_tensor(f) else f.cpu().numpy() for _, _, f in items], 0), None, None)
            bi = torch.tensor([b for b, _, _ in items], device=self.device)
            si = torch.tensor([s_ for _, s_, _ in items], device=self.device)
            j, o = self.prep(fr, pd[bi], h, w)
            img[bi, si], orig[bi, si] = j, o
        out["img"], out["img_orig"] = img[:, 0].contiguous(), orig[:, 0].contiguous()
        if nctx:
            out["ctx_img"] = [img[:, 1 + i].contiguous() for i in range(nctx)]
            out["ctx_img_orig"] = [orig[:, 1 + i].contiguous() for i in range(nctx)]
        return out
