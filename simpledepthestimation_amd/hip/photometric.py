"""Autograd wrappers over the photometric-path entry points of libsde_hip.so (include/sde_hip.h).

Each Function owns the tensors its backward needs; kernels never allocate.  Everything is enqueued on
torch's current stream and never synchronises (hipGraph-capture safe).
"""
import ctypes

import torch

from . import lib as L


def _f32c(t):
    if t.dtype != torch.float32:
        raise L.SdeHipError(f"expected float32 tensor, got {t.dtype}")
    return t.contiguous()


def resize(image, size, mode="bilinear"):
    """camera.py:L40-46 resize_img (no autograd: only images / ground truth are resized on this path)."""
    image = _f32c(image)
    B, C, H, W = image.shape
    h, w = int(size[0]), int(size[1])
    if (H, W) == (h, w):
        return image
    out = torch.empty(B, C, h, w, device=image.device, dtype=torch.float32)
    L.check(L.lib().sde_resize(L.ptr(image), L.ptr(out), B * C, H, W, h, w, 0 if mode == "bilinear" else 1, L.stream()), "sde_resize")
    return out


def view_synthesis_raw(image_B, depth_A, K, pose, sx=1.0, sy=1.0, want_indices=True):
    """camera.py:L166-202 forward only.  Returns dict(sampled, Z, grid, valid, fx, fy)."""
    image_B, depth_A, K, pose = _f32c(image_B), _f32c(depth_A), _f32c(K), _f32c(pose)
    B, C, H, W = image_B.shape
    dev = image_B.device
    out = {"sampled": torch.empty(B, C, H, W, device=dev), "Z": torch.empty(B, 1, H, W, device=dev),
           "grid": torch.empty(B, H, W, 2, device=dev), "valid": torch.empty(B, 1, H, W, device=dev, dtype=torch.uint8)}
    if want_indices:
        out["fx"] = torch.empty(B, H, W, device=dev, dtype=torch.int32)
        out["fy"] = torch.empty(B, H, W, device=dev, dtype=torch.int32)
    L.check(L.lib().sde_view_synthesis(L.ptr(image_B), L.ptr(depth_A), L.ptr(K), L.ptr(pose), sx, sy, B, C, H, W, L.ptr(out["sampled"]),
                                       L.ptr(out["Z"]), L.ptr(out["grid"]), L.ptr(out["valid"]), L.ptr(out.get("fx")), L.ptr(out.get("fy")),
                                       L.stream()), "sde_view_synthesis")
    return out


class _PoseVec2Mat(torch.autograd.Function):
    @staticmethod
    def forward(ctx, vec):
        vec = _f32c(vec)
        n = vec.shape[0]
        mat = torch.empty(n, 4, 4, device=vec.device)
        L.check(L.lib().sde_pose_vec2mat(L.ptr(vec), L.ptr(mat), n, L.stream()), "sde_pose_vec2mat")
        ctx.save_for_backward(vec)
        return mat

    @staticmethod
    def backward(ctx, dmat):
        (vec,) = ctx.saved_tensors
        dvec = torch.empty_like(vec)
        L.check(L.lib().sde_pose_vec2mat_bwd(L.ptr(vec), L.ptr(_f32c(dmat)), L.ptr(dvec), vec.shape[0], L.stream()), "sde_pose_vec2mat_bwd")
        return dvec


def pose_vec2mat(vec):
    """pose_utils.py:L130-137."""
    return _PoseVec2Mat.apply(vec)


def _desc(A, ctxs, poses, depth, K, sx, sy, ssim_w, C1, C2, automask, reduce_mean, clip_thr=None):
    d = L.PhotoDesc()
    d.clip_thr = clip_thr.data_ptr() if clip_thr is not None else 0
    B, _, h, w = depth.shape
    d.A = A.data_ptr(); d.depth = depth.data_ptr(); d.K = K.data_ptr()
    for j, (c, p) in enumerate(zip(ctxs, poses)):
        d.ctx[j] = c.data_ptr(); d.pose[j] = p.data_ptr()
    d.B, d.h, d.w, d.nctx = B, h, w, len(ctxs)
    d.automask, d.reduce_mean = int(bool(automask)), int(bool(reduce_mean))
    d.sx, d.sy, d.ssim_w, d.C1, d.C2 = sx, sy, ssim_w, C1, C2
    return d


class _PhotoScale(torch.autograd.Function):
    """One scale of MonoDepth2's photometric loss (MonoDepth2.py:L78-101,L116-124): returns mean(reduced map)."""

    @staticmethod
    def forward(ctx, depth, K, A, sx, sy, ssim_w, C1, C2, automask, reduce_mean, clip, nctx, *rest):
        ctxs, poses = rest[:nctx], rest[nctx:]
        depth, K, A = _f32c(depth), _f32c(K), _f32c(A)
        ctxs = [_f32c(c) for c in ctxs]
        poses = [_f32c(p) for p in poses]
        B, _, h, w = depth.shape
        dev = depth.device
        lib = L.lib()
        thr = None
        if clip > 0.0:
            # LOSS.CLIP (MonoDepth2.py:L147-149): every map is clamped at mean + clip * std of ITSELF; one extra forward yields the unclipped
            # maps, the per-map statistics stay on the device (the reference converts them to a python float: same fp32 value, no gradient)
            maps = photometric_maps(depth, K, A, ctxs, poses, sx, sy, ssim_w, C1, C2, automask, "mean" if reduce_mean else "min")["maps"]
            thr = (maps.mean((0, 2, 3)) + clip * maps.std((0, 2, 3))).contiguous()
        d = _desc(A, ctxs, poses, depth, K, sx, sy, ssim_w, C1, C2, automask, reduce_mean, thr)
        sampled = [torch.empty(B, 3, h, w, device=dev) for _ in range(nctx)]
        sel = torch.empty(B, h, w, device=dev, dtype=torch.uint8)
        partial = torch.empty(lib.sde_photo_num_blocks(B, h, w, 0), device=dev)
        loss = torch.empty((), device=dev)
        # algorithmic bytes (fused per scale): target 12 + contexts 12 each + depth 4 per pixel, fp32
        L.timed("photo_fwd", B * h * w * (16 + 12 * nctx), nctx,
                lambda: L.check(lib.sde_photo_fwd(ctypes.byref(d), L.ptr_array(sampled), L.ptr(sel), None, L.ptr(partial), L.ptr(loss), 1.0, 0, L.stream()),
                                "sde_photo_fwd"), dict(B=B, h=h, w=w, moved=B * h * w * (16 + 24 * nctx + 1)))
        ctx.save_for_backward(depth, K, A, sel, *ctxs, *poses, *sampled)
        ctx.cfg = (sx, sy, ssim_w, C1, C2, automask, reduce_mean, nctx)
        ctx.thr = thr
        return loss

    @staticmethod
    def backward(ctx, gout):
        sx, sy, ssim_w, C1, C2, automask, reduce_mean, nctx = ctx.cfg
        t = ctx.saved_tensors
        depth, K, A, sel = t[:4]
        ctxs, poses, sampled = t[4:4 + nctx], t[4 + nctx:4 + 2 * nctx], t[4 + 2 * nctx:4 + 3 * nctx]
        B, _, h, w = depth.shape
        dev = depth.device
        lib = L.lib()
        d = _desc(A, ctxs, poses, depth, K, sx, sy, ssim_w, C1, C2, automask, reduce_mean, ctx.thr)
        d_depth = torch.empty_like(depth)
        d_pose = [torch.empty(B, 4, 4, device=dev) for _ in range(nctx)]
        pp = torch.empty(lib.sde_photo_num_blocks(B, h, w, 1) * nctx * 12, device=dev)
        gout = _f32c(gout)
        # algorithmic bytes: target 12 + (context 12 + saved warped frame 12) per context + depth 4 + arg-min 1 + d_depth 4 per pixel
        L.timed("photo_bwd", B * h * w * (21 + 24 * nctx), nctx,
                lambda: L.check(lib.sde_photo_bwd(ctypes.byref(d), L.ptr_array(sampled), L.ptr(sel), L.ptr(gout), 1.0, L.ptr(d_depth), 0, L.ptr(pp),
                                                  L.ptr_array(d_pose), 0, L.stream()), "sde_photo_bwd"), dict(B=B, h=h, w=w))
        return (d_depth, None, None, None, None, None, None, None, None, None, None, None) + (None,) * nctx + tuple(d_pose)


def photometric_scale_loss(depth, K, A, ctxs, poses, sx, sy, ssim_w=0.85, C1=1e-4, C2=9e-4, automask=True, reduce="min", clip=0.0):
    if reduce not in ("min", "mean"):
        raise NotImplementedError(reduce)           # MonoDepth2.py:L121-122
    return _PhotoScale.apply(depth, K, A, float(sx), float(sy), float(ssim_w), float(C1), float(C2), bool(automask), reduce == "mean",
                             float(clip), len(ctxs), *ctxs, *poses)


PH_MAX_SCALES = 4     # PH_MAX_SCALES of csrc/photometric.hip


class _PhotoMulti(torch.autograd.Function):
    """Every scale of MonoDepth2's photometric loss (MonoDepth2.py:L78-112) in one launch per phase: returns the [n] per-scale means."""

    @staticmethod
    def forward(ctx, K, ssim_w, C1, C2, automask, reduce_mean, n, nctx, scales, pose_stream, *rest):
        depths, As = rest[:n], rest[n:2 * n]
        ctxs = rest[2 * n:2 * n + n * nctx]               # scale-major
        poses = rest[2 * n + n * nctx:]
        K = _f32c(K)
        depths = [_f32c(d) for d in depths]; As = [_f32c(a) for a in As]
        ctxs = [_f32c(c) for c in ctxs]; poses = [_f32c(p) for p in poses]
        dev = K.device
        lib = L.lib()
        B = depths[0].shape[0]
        descs = (L.PhotoDesc * n)()
        sampled, sels, partials = [], [], []
        samp_arr = (ctypes.c_void_p * (n * L.MAX_CTX))()
        sel_arr, part_arr = (ctypes.c_void_p * n)(), (ctypes.c_void_p * n)()
        for s in range(n):
            _, _, h, w = depths[s].shape
            sx, sy = scales[s]
            d = _desc(As[s], ctxs[s * nctx:(s + 1) * nctx], poses, depths[s], K, sx, sy, ssim_w, C1, C2, automask, reduce_mean)
            ctypes.memmove(ctypes.byref(descs, s * ctypes.sizeof(L.PhotoDesc)), ctypes.byref(d), ctypes.sizeof(L.PhotoDesc))
            sm = [torch.empty(B, 3, h, w, device=dev) for _ in range(nctx)]
            for j, t in enumerate(sm):
                samp_arr[s * L.MAX_CTX + j] = t.data_ptr()
            sampled += sm
            sels.append(torch.empty(B, h, w, device=dev, dtype=torch.uint8)); sel_arr[s] = sels[-1].data_ptr()
            partials.append(torch.empty(lib.sde_photo_num_blocks(B, h, w, 0), device=dev)); part_arr[s] = partials[-1].data_ptr()
        loss = torch.empty(n, device=dev)
        h0, w0 = depths[0].shape[-2:]
        nbytes = sum(B * d.shape[-2] * d.shape[-1] * (16 + 12 * nctx) for d in depths)
        L.timed("photo_fwd", nbytes, nctx, lambda: L.check(lib.sde_photo_multi_fwd(descs, n, samp_arr, sel_arr, part_arr, L.ptr(loss), L.stream()), "sde_photo_multi_fwd"),
                dict(B=B, h=h0, w=w0, scales=n))
        ctx.save_for_backward(K, *depths, *As, *ctxs, *poses, *sampled, *sels)
        ctx.cfg = (ssim_w, C1, C2, automask, reduce_mean, n, nctx, scales)
        ctx.pose_stream = pose_stream
        return loss

    @staticmethod
    def backward(ctx, gout):
        ssim_w, C1, C2, automask, reduce_mean, n, nctx, scales = ctx.cfg
        t = ctx.saved_tensors
        K = t[0]
        o = 1
        depths = t[o:o + n]; o += n
        As = t[o:o + n]; o += n
        ctxs = t[o:o + n * nctx]; o += n * nctx
        poses = t[o:o + nctx]; o += nctx
        sampled = t[o:o + n * nctx]; o += n * nctx
        sels = t[o:o + n]
        dev = K.device
        lib = L.lib()
        B = depths[0].shape[0]
        descs = (L.PhotoDesc * n)()
        samp_arr = (ctypes.c_void_p * (n * L.MAX_CTX))()
        sel_arr, dd_arr, pp_arr = (ctypes.c_void_p * n)(), (ctypes.c_void_p * n)(), (ctypes.c_void_p * n)()
        d_depths, pps = [], []
        for s in range(n):
            _, _, h, w = depths[s].shape
            sx, sy = scales[s]
            d = _desc(As[s], ctxs[s * nctx:(s + 1) * nctx], poses, depths[s], K, sx, sy, ssim_w, C1, C2, automask, reduce_mean)
            ctypes.memmove(ctypes.byref(descs, s * ctypes.sizeof(L.PhotoDesc)), ctypes.byref(d), ctypes.sizeof(L.PhotoDesc))
            for j in range(nctx):
                samp_arr[s * L.MAX_CTX + j] = sampled[s * nctx + j].data_ptr()
            sel_arr[s] = sels[s].data_ptr()
            d_depths.append(torch.empty_like(depths[s])); dd_arr[s] = d_depths[-1].data_ptr()
            pps.append(torch.empty(lib.sde_photo_num_blocks(B, h, w, 1) * nctx * 12, device=dev)); pp_arr[s] = pps[-1].data_ptr()
        d_pose = [torch.empty(B, 4, 4, device=dev) for _ in range(nctx)]
        gout = _f32c(gout)
        h0, w0 = depths[0].shape[-2:]
        nbytes = sum(B * d.shape[-2] * d.shape[-1] * (21 + 24 * nctx) for d in depths)
        # PoseNet on the auxiliary stream (MonoDepth2Model.forward): the pose side of this backward -- summing the partials -- goes there as well, and is
        # enqueued BEFORE the main stream's next kernel.  Besides taking a launch off the main chain this shapes the captured graph: the HIP runtime gives
        # a node's first captured dependant the node's own queue and the later ones the next queues round robin, and with the pose chain as the SECOND
        # dependant it landed on the queue of the depth network's data-gradient chain and ran after it (in-graph markers, profiles/README.md round 3)
        # (pose_stream is the caller's statement that every consumer of the pose gradients runs on that stream)
        aux = ctx.pose_stream if L.PROFILE is None else None
        L.timed("photo_bwd", nbytes, nctx, lambda: L.check(lib.sde_photo_multi_bwd(descs, n, samp_arr, sel_arr, L.ptr(gout), dd_arr, pp_arr,
                                                                                   None if aux is not None else L.ptr_array(d_pose), L.stream()),
                                                          "sde_photo_multi_bwd"), dict(B=B, h=h0, w=w0, scales=n))
        if aux is not None:
            aux.wait_stream(torch.cuda.current_stream())
            with torch.cuda.stream(aux):
                L.check(lib.sde_photo_multi_pose_finalize(descs, n, pp_arr, L.ptr_array(d_pose), L.stream()), "sde_photo_multi_pose_finalize")
            for t_ in pps + d_pose:
                t_.record_stream(aux)
        return (None,) * 10 + tuple(d_depths) + (None,) * n + (None,) * (n * nctx) + tuple(d_pose)


def photometric_multi_loss(depths, K, As, ctxs_per_scale, poses, scales, ssim_w=0.85, C1=1e-4, C2=9e-4, automask=True, reduce="min", pose_stream=None):
    """All scales in one launch per phase.  depths / As: per scale; ctxs_per_scale[s]: the context frames at scale s; scales[s] = (w_s / W, h_s / H).
    Returns a [n] tensor: photometric_scale_loss of every scale (no LOSS.CLIP: that option keeps the per-scale path).
    pose_stream: the stream the network that produced `poses` ran on when that is not the current one (its backward runs there): the pose gradients
    are then summed on that stream."""
    if reduce not in ("min", "mean"):
        raise NotImplementedError(reduce)
    n, nctx = len(depths), len(poses)
    if not (1 <= n <= PH_MAX_SCALES and len(As) == n and len(ctxs_per_scale) == n and all(len(c) == nctx for c in ctxs_per_scale)):
        raise L.SdeHipError("photometric_multi_loss: 1-4 scales, one target and nctx context frames per scale")
    flat_ctx = [c for cs in ctxs_per_scale for c in cs]
    return _PhotoMulti.apply(K, float(ssim_w), float(C1), float(C2), bool(automask), reduce == "mean", n, nctx, tuple((float(a), float(b)) for a, b in scales),
                             pose_stream, *depths, *As, *flat_ctx, *poses)


_TICKET = {}


def _ticket(dev):
    """The finalize kernel's arrival counter: one device int per device, zero between launches (the kernel resets it)."""
    if dev not in _TICKET:
        _TICKET[dev] = torch.zeros(1, dtype=torch.int32, device=dev)
    return _TICKET[dev]


class _MonoLoss(torch.autograd.Function):
    """Photometric + smoothness terms of every scale (MonoDepth2.py:L78-126): (rec_loss, smooth_loss, per-scale values [2n]) in four launches forward and
    two backward; the depth gradient of a scale is written once (photometric) and accumulated once (smoothness) instead of being summed by autograd."""

    @staticmethod
    def forward(ctx, K, ssim_w, C1, C2, automask, reduce_mean, n, nctx, scales, photo_w, smooth_w, pose_stream, *rest):
        depths, As = rest[:n], rest[n:2 * n]
        ctxs = rest[2 * n:2 * n + n * nctx]               # scale-major
        poses = rest[2 * n + n * nctx:]
        K = _f32c(K)
        depths = [_f32c(d) for d in depths]; As = [_f32c(a) for a in As]
        ctxs = [_f32c(c) for c in ctxs]; poses = [_f32c(p) for p in poses]
        dev = K.device
        lib = L.lib()
        B = depths[0].shape[0]
        descs = (L.PhotoDesc * n)()
        sampled, sels, partials, sm = [], [], [], []
        samp_arr = (ctypes.c_void_p * (n * L.MAX_CTX))()
        arr = lambda: (ctypes.c_void_p * n)()
        sel_arr, part_arr, mean_arr, dn_arr, lp_arr, sp_arr = arr(), arr(), arr(), arr(), arr(), arr()
        for s in range(n):
            _, _, h, w = depths[s].shape
            sx, sy = scales[s]
            d = _desc(As[s], ctxs[s * nctx:(s + 1) * nctx], poses, depths[s], K, sx, sy, ssim_w, C1, C2, automask, reduce_mean)
            ctypes.memmove(ctypes.byref(descs, s * ctypes.sizeof(L.PhotoDesc)), ctypes.byref(d), ctypes.sizeof(L.PhotoDesc))
            smp = [torch.empty(B, 3, h, w, device=dev) for _ in range(nctx)]
            for j, t in enumerate(smp):
                samp_arr[s * L.MAX_CTX + j] = t.data_ptr()
            sampled += smp
            sels.append(torch.empty(B, h, w, device=dev, dtype=torch.uint8)); sel_arr[s] = sels[-1].data_ptr()
            partials.append(torch.empty(lib.sde_photo_num_blocks(B, h, w, 0), device=dev)); part_arr[s] = partials[-1].data_ptr()
            if smooth_w is not None:
                nb = lib.sde_smooth_num_blocks(B, h, w)
                mean_part, dn, loss_part, s_part = torch.empty(B * 32, device=dev), torch.empty(B, h, w, device=dev), torch.empty(nb, device=dev), torch.empty(nb, device=dev)
                sm += [mean_part, dn, s_part]
                mean_arr[s], dn_arr[s], lp_arr[s], sp_arr[s] = mean_part.data_ptr(), dn.data_ptr(), loss_part.data_ptr(), s_part.data_ptr()
                partials.append(loss_part)
        per_scale = torch.empty(2 * n, device=dev)
        totals = torch.empty(2, device=dev)
        pw = (ctypes.c_float * n)(*photo_w)
        sw = (ctypes.c_float * n)(*smooth_w) if smooth_w is not None else None
        h0, w0 = depths[0].shape[-2:]
        nbytes = sum(B * d.shape[-2] * d.shape[-1] * (16 + 12 * nctx) for d in depths)
        L.timed("photo_fwd", nbytes, nctx, lambda: L.check(lib.sde_mono_loss_fwd(descs, n, pw, sw, samp_arr, sel_arr, part_arr, mean_arr, dn_arr, lp_arr, sp_arr,
                                                                                 L.ptr(per_scale), L.ptr(totals), L.ptr(_ticket(dev)), L.stream()), "sde_mono_loss_fwd"),
                dict(B=B, h=h0, w=w0, scales=n))
        ctx.save_for_backward(K, *depths, *As, *ctxs, *poses, *sampled, *sels, *sm)
        ctx.cfg = (ssim_w, C1, C2, automask, reduce_mean, n, nctx, scales, photo_w, smooth_w)
        ctx.pose_stream = pose_stream
        ctx.mark_non_differentiable(per_scale)
        return totals[0], totals[1], per_scale

    @staticmethod
    def backward(ctx, g_rec, g_smooth, _g_per_scale):
        ssim_w, C1, C2, automask, reduce_mean, n, nctx, scales, photo_w, smooth_w = ctx.cfg
        t = ctx.saved_tensors
        K = t[0]
        o = 1
        depths = t[o:o + n]; o += n
        As = t[o:o + n]; o += n
        ctxs = t[o:o + n * nctx]; o += n * nctx
        poses = t[o:o + nctx]; o += nctx
        sampled = t[o:o + n * nctx]; o += n * nctx
        sels = t[o:o + n]; o += n
        sm = t[o:]
        dev = K.device
        lib = L.lib()
        B = depths[0].shape[0]
        descs = (L.PhotoDesc * n)()
        samp_arr = (ctypes.c_void_p * (n * L.MAX_CTX))()
        arr = lambda: (ctypes.c_void_p * n)()
        sel_arr, dd_arr, pp_arr, mean_arr, dn_arr, sp_arr = arr(), arr(), arr(), arr(), arr(), arr()
        d_depths, pps = [], []
        for s in range(n):
            _, _, h, w = depths[s].shape
            sx, sy = scales[s]
            d = _desc(As[s], ctxs[s * nctx:(s + 1) * nctx], poses, depths[s], K, sx, sy, ssim_w, C1, C2, automask, reduce_mean)
            ctypes.memmove(ctypes.byref(descs, s * ctypes.sizeof(L.PhotoDesc)), ctypes.byref(d), ctypes.sizeof(L.PhotoDesc))
            for j in range(nctx):
                samp_arr[s * L.MAX_CTX + j] = sampled[s * nctx + j].data_ptr()
            sel_arr[s] = sels[s].data_ptr()
            d_depths.append(torch.empty_like(depths[s])); dd_arr[s] = d_depths[-1].data_ptr()
            pps.append(torch.empty(lib.sde_photo_num_blocks(B, h, w, 1) * nctx * 12, device=dev)); pp_arr[s] = pps[-1].data_ptr()
            if smooth_w is not None:
                mean_arr[s], dn_arr[s], sp_arr[s] = sm[3 * s].data_ptr(), sm[3 * s + 1].data_ptr(), sm[3 * s + 2].data_ptr()
        d_pose = [torch.empty(B, 4, 4, device=dev) for _ in range(nctx)]
        if g_rec is None:                       # rec_loss not part of the objective: its gradient is zero
            g_rec = torch.zeros((), device=dev)
        g_rec = _f32c(g_rec)
        g_smooth = _f32c(g_smooth) if (g_smooth is not None and smooth_w is not None) else None
        pw = (ctypes.c_float * n)(*photo_w)
        sw = (ctypes.c_float * n)(*smooth_w) if smooth_w is not None else None
        h0, w0 = depths[0].shape[-2:]
        nbytes = sum(B * d.shape[-2] * d.shape[-1] * (21 + 24 * nctx) for d in depths)
        # the pose side goes to PoseNet's stream, enqueued before the main stream's next kernel (see _PhotoMulti.backward)
        aux = ctx.pose_stream if L.PROFILE is None else None
        L.timed("photo_bwd", nbytes, nctx, lambda: L.check(lib.sde_mono_loss_bwd(descs, n, pw, None, samp_arr, sel_arr, L.ptr(g_rec), None, None, None, None, dd_arr, pp_arr,
                                                                                 None if aux is not None else L.ptr_array(d_pose), L.stream()), "sde_mono_loss_bwd"),
                dict(B=B, h=h0, w=w0, scales=n))
        if aux is not None:
            aux.wait_stream(torch.cuda.current_stream())
            with torch.cuda.stream(aux):
                L.check(lib.sde_photo_multi_pose_finalize(descs, n, pp_arr, L.ptr_array(d_pose), L.stream()), "sde_photo_multi_pose_finalize")
            for t_ in pps + d_pose:
                t_.record_stream(aux)
        if g_smooth is not None:
            L.check(lib.sde_smooth_multi_bwd(descs, n, sw, L.ptr(g_smooth), mean_arr, dn_arr, sp_arr, dd_arr, 1, L.stream()), "sde_smooth_multi_bwd")
        return (None,) * 12 + tuple(d_depths) + (None,) * n + (None,) * (n * nctx) + tuple(d_pose)


def mono_loss(depths, K, As, ctxs_per_scale, poses, scales, photo_w, smooth_w=None, ssim_w=0.85, C1=1e-4, C2=9e-4, automask=True, reduce="min", pose_stream=None):
    """MonoDepth2's loss loop over the scales (MonoDepth2.py:L78-126), photometric and smoothness terms: returns (rec_loss, smooth_loss, per_scale) with
    rec_loss = sum_s photo_w[s] * photometric_scale_loss(scale s), smooth_loss = sum_s smooth_w[s] * smoothness_loss(scale s) (zero without smooth_w) and
    per_scale [2n] the unweighted terms (no gradient).  Arguments as for photometric_multi_loss."""
    if reduce not in ("min", "mean"):
        raise NotImplementedError(reduce)
    n, nctx = len(depths), len(poses)
    if not (1 <= n <= PH_MAX_SCALES and len(As) == n and len(ctxs_per_scale) == n and all(len(c) == nctx for c in ctxs_per_scale) and len(photo_w) == n
            and (smooth_w is None or len(smooth_w) == n)):
        raise L.SdeHipError("mono_loss: 1-4 scales, one target, nctx context frames and one weight per scale")
    flat_ctx = [c for cs in ctxs_per_scale for c in cs]
    return _MonoLoss.apply(K, float(ssim_w), float(C1), float(C2), bool(automask), reduce == "mean", n, nctx, tuple((float(a), float(b)) for a, b in scales),
                           tuple(float(v) for v in photo_w), None if smooth_w is None else tuple(float(v) for v in smooth_w), pose_stream,
                           *depths, *As, *flat_ctx, *poses)


def photometric_maps(depth, K, A, ctxs, poses, sx, sy, ssim_w=0.85, C1=1e-4, C2=9e-4, automask=True, reduce="min"):
    """Forward only, exposing the individual maps / sampled frames / arg-min (tests, debugging)."""
    depth, K, A = _f32c(depth), _f32c(K), _f32c(A)
    ctxs = [_f32c(c) for c in ctxs]; poses = [_f32c(p) for p in poses]
    B, _, h, w = depth.shape
    dev = depth.device
    lib = L.lib()
    nctx = len(ctxs)
    d = _desc(A, ctxs, poses, depth, K, sx, sy, ssim_w, C1, C2, automask, reduce == "mean")
    nmaps = 2 * nctx if automask else nctx
    sampled = [torch.empty(B, 3, h, w, device=dev) for _ in range(nctx)]
    sel = torch.empty(B, h, w, device=dev, dtype=torch.uint8)
    maps = torch.empty(B, nmaps, h, w, device=dev)
    partial = torch.empty(lib.sde_photo_num_blocks(B, h, w, 0), device=dev)
    loss = torch.empty((), device=dev)
    L.check(lib.sde_photo_fwd(ctypes.byref(d), L.ptr_array(sampled), L.ptr(sel), L.ptr(maps), L.ptr(partial), L.ptr(loss), 1.0, 0, L.stream()),
            "sde_photo_fwd")
    return {"loss": loss, "maps": maps, "sampled": sampled, "sel": sel}


class _Smooth(torch.autograd.Function):
    @staticmethod
    def forward(ctx, depth, image):
        depth, image = _f32c(depth), _f32c(image)
        B, _, h, w = depth.shape
        dev = depth.device
        lib = L.lib()
        nb = lib.sde_smooth_num_blocks(B, h, w)
        mean_part = torch.empty(B * 32, device=dev)
        dn = torch.empty(B, h, w, device=dev)
        loss_part = torch.empty(nb, device=dev)
        s_part = torch.empty(nb, device=dev)
        loss = torch.empty((), device=dev)
        L.check(lib.sde_smooth_fwd(L.ptr(depth), L.ptr(image), B, h, w, L.ptr(mean_part), L.ptr(dn), L.ptr(loss_part), L.ptr(s_part), L.ptr(loss),
                                   1.0, 0, L.stream()), "sde_smooth_fwd")
        ctx.save_for_backward(depth, dn, mean_part, s_part)
        return loss

    @staticmethod
    def backward(ctx, gout):
        depth, dn, mean_part, s_part = ctx.saved_tensors
        B, _, h, w = depth.shape
        d_depth = torch.empty_like(depth)
        L.check(L.lib().sde_smooth_bwd(L.ptr(depth), L.ptr(dn), L.ptr(mean_part), L.ptr(s_part), L.ptr(_f32c(gout)), 1.0, B, h, w, L.ptr(d_depth),
                                       0, L.stream()), "sde_smooth_bwd")
        return d_depth, None


def smoothness_loss(depth, image):
    """smoothness_loss.py:L42-80."""
    return _Smooth.apply(depth, image)


class _SSIMMap(torch.autograd.Function):
    """The stand-alone SSIM distance map (ssim_loss.py:L34-53) with gradients to both images."""

    @staticmethod
    def forward(ctx, x, y, C1, C2):
        x, y = _f32c(x), _f32c(y)
        if x.shape != y.shape or x.dim() != 4:
            raise L.SdeHipError(f"SSIM: x and y must be [B,C,H,W] of one shape, got {tuple(x.shape)} and {tuple(y.shape)}")
        B, C, H, W = x.shape
        out = torch.empty_like(x)
        L.check(L.lib().sde_ssim_fwd(L.ptr(x), L.ptr(y), B, C, H, W, C1, C2, L.ptr(out), L.stream()), "sde_ssim_fwd")
        ctx.save_for_backward(x, y)
        ctx.cfg = (C1, C2)
        return out

    @staticmethod
    def backward(ctx, gout):
        x, y = ctx.saved_tensors
        B, C, H, W = x.shape
        C1, C2 = ctx.cfg
        dx = torch.empty_like(x) if ctx.needs_input_grad[0] else None
        dy = torch.empty_like(y) if ctx.needs_input_grad[1] else None
        if dx is None and dy is None:
            return None, None, None, None
        ws = torch.empty(B * C * H * W, 4, device=x.device)
        L.check(L.lib().sde_ssim_bwd(L.ptr(x), L.ptr(y), L.ptr(_f32c(gout)), B, C, H, W, C1, C2, L.ptr(ws), L.ptr(dx), L.ptr(dy), L.stream()), "sde_ssim_bwd")
        return dx, dy, None, None


def ssim_map(x, y, C1=1e-4, C2=9e-4):
    """clamp((1 - SSIM(x, y)) / 2, 0, 1) per pixel and channel, [B,C,H,W] fp32 (ssim_loss.py:L34-53)."""
    return _SSIMMap.apply(x, y, float(C1), float(C2))


class _Silog(torch.autograd.Function):
    @staticmethod
    def forward(ctx, est, gt, vf):
        est, gt = _f32c(est), _f32c(gt)
        B, _, h, w = est.shape
        H, W = gt.shape[-2:]
        dev = est.device
        lib = L.lib()
        part = torch.empty(lib.sde_silog_num_blocks(B, h, w) * 3, device=dev)
        stats = torch.empty(4, device=dev)
        L.check(lib.sde_silog_fwd(L.ptr(est), L.ptr(gt), B, h, w, H, W, vf, L.ptr(part), L.ptr(stats), L.stream()), "sde_silog_fwd")
        ctx.save_for_backward(est, gt, stats)
        ctx.vf = vf
        return stats[3].clone()

    @staticmethod
    def backward(ctx, gout):
        est, gt, stats = ctx.saved_tensors
        B, _, h, w = est.shape
        H, W = gt.shape[-2:]
        d_est = torch.empty_like(est)
        L.check(L.lib().sde_silog_bwd(L.ptr(est), L.ptr(gt), L.ptr(stats), L.ptr(_f32c(gout)), 1.0, ctx.vf, B, h, w, H, W, L.ptr(d_est), 0,
                                      L.stream()), "sde_silog_bwd")
        return d_est, None, None


class _SilogMulti(torch.autograd.Function):
    """sum_k weights[k] * SILog(ests[k], nearest(gt)) over the prediction scales: one forward launch + one finalize, one backward launch."""

    @staticmethod
    def forward(ctx, gt, vf, weights, *ests):
        import ctypes
        ests = [_f32c(e) for e in ests]
        gt = _f32c(gt)
        n, B = len(ests), ests[0].shape[0]
        H, W = gt.shape[-2:]
        dev = gt.device
        lib = L.lib()
        hs = (ctypes.c_int * n)(*[e.shape[-2] for e in ests])
        ws = (ctypes.c_int * n)(*[e.shape[-1] for e in ests])
        wt = (ctypes.c_float * n)(*[float(x) for x in weights])
        ep = (ctypes.c_void_p * n)(*[e.data_ptr() for e in ests])
        part = torch.empty(lib.sde_silog_multi_num_blocks(B, hs, ws, n) * 3, device=dev)
        stats = torch.empty(n, 4, device=dev)
        total = torch.empty((), device=dev)
        L.check(lib.sde_silog_multi_fwd(ep, L.ptr(gt), B, hs, ws, wt, n, H, W, vf, L.ptr(part), L.ptr(stats), L.ptr(total), L.stream()), "sde_silog_multi_fwd")
        ctx.save_for_backward(gt, stats, *ests)
        ctx.cfg = (vf, tuple(float(x) for x in weights))
        return total

    @staticmethod
    def backward(ctx, gout):
        import ctypes
        gt, stats, *ests = ctx.saved_tensors
        vf, weights = ctx.cfg
        n, B = len(ests), ests[0].shape[0]
        H, W = gt.shape[-2:]
        d = [torch.empty_like(e) for e in ests]
        hs = (ctypes.c_int * n)(*[e.shape[-2] for e in ests])
        ws = (ctypes.c_int * n)(*[e.shape[-1] for e in ests])
        wt = (ctypes.c_float * n)(*weights)
        ep = (ctypes.c_void_p * n)(*[e.data_ptr() for e in ests])
        dp = (ctypes.c_void_p * n)(*[x.data_ptr() for x in d])
        L.check(L.lib().sde_silog_multi_bwd(ep, L.ptr(gt), L.ptr(stats), L.ptr(_f32c(gout)), 1.0, vf, B, hs, ws, wt, n, H, W, dp, L.stream()), "sde_silog_multi_bwd")
        return (None, None, None) + tuple(d)


MAX_SILOG_SCALES = 4


def silog_loss_multi(depth_ests, depth_gt_full, variance_focus, weights):
    """sum_k weights[k] * silog_loss(depth_ests[k], depth_gt_full) (Supervised.py:L42-47) in one launch per phase."""
    if not 1 <= len(depth_ests) <= MAX_SILOG_SCALES or len(weights) != len(depth_ests):
        raise L.SdeHipError(f"silog_loss_multi: 1..{MAX_SILOG_SCALES} scales with one weight each, got {len(depth_ests)} / {len(weights)}")
    return _SilogMulti.apply(depth_gt_full, float(variance_focus), tuple(weights), *depth_ests)


def silog_loss(depth_est, depth_gt_full, variance_focus=0.85):
    """losses.py:L10-13 applied to (pred, resize_img(gt, pred.shape, 'nearest')) -- Supervised.py:L44-45 -- in one kernel."""
    return _Silog.apply(depth_est, depth_gt_full, float(variance_focus))
