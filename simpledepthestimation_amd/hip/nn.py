"""Autograd wrappers over the convolution engine and its surrounding layers (include/sde_hip.h).

Tensor convention: activations are NHWC torch tensors [B, H, W, C] (float32, bfloat16 or float16), C padded to 16 bytes;
parameters stay fp32 in the reference's layouts (conv weight OIHW) so state dicts are interchangeable.
"""
import ctypes
import weakref
import os
from ctypes import POINTER, Structure, c_float, c_int, c_int32, c_long, c_void_p

import torch

from . import lib as L

SRC_PLAIN, SRC_UPCAT, SRC_ZEROINS = 0, 1, 2
REDUCE_ROWS = 32   # SDE_REDUCE_ROWS: scratch rows every partial-sum slab carries behind its payload
ACT_NONE, ACT_ELU, ACT_RELU = 0, 1, 2


class ConvDesc(Structure):
    _fields_ = [("x0", c_void_p), ("x1", c_void_p), ("dtype", c_int32), ("C0", c_int32), ("C1", c_int32), ("H0", c_int32), ("W0", c_int32),
                ("IH", c_int32), ("IW", c_int32), ("src_mode", c_int32), ("KH", c_int32), ("KW", c_int32), ("stride", c_int32), ("pad", c_int32),
                ("reflect", c_int32), ("Bn", c_int32), ("OH", c_int32), ("OW", c_int32)]


_P, _I, _F, _LG = c_void_p, c_int, c_float, c_long
L.register_protos({
    "sde_pack_weight": ([_P, _P, _I, _I, _I, _I, _I, _I, _I, _I, _P], c_int),
    "sde_pack_weights_batched": ([_P, _I, _LG, _I, _P], c_int),
    "sde_pack_item_blocks": ([_I, _I, _I, _I], c_int),
    "sde_conv_fwd": ([POINTER(ConvDesc), _P, _P, _I, _P, _I, _I, _P, _P], c_int),
    "sde_conv_fwd_tiles_m": ([POINTER(ConvDesc), _I], c_int),
    "sde_conv_fwd_ws_bytes": ([POINTER(ConvDesc), _I], ctypes.c_size_t),
    "sde_conv_fwd_ws": ([POINTER(ConvDesc), _P, _P, _I, _P, _I, _I, _P, _P, ctypes.c_size_t, _P], c_int),
    "sde_conv_fwd_variant": ([POINTER(ConvDesc), _I], c_int),
    "sde_conv_dgrad_bnbwd_rows": ([POINTER(ConvDesc), _I, _I], c_int),
    "sde_conv_dgrad_bnbwd": ([POINTER(ConvDesc), _P, _P, _I, _I, _P, _P, _P, _P], c_int),
    "sde_conv_dgrad_bnbwd_res_rows": ([POINTER(ConvDesc), _I, _I], c_int),
    "sde_conv_dgrad_bnbwd_res": ([POINTER(ConvDesc), _P, _P, _I, _I, _P, _P, _P, _P, _P, _P], c_int),
    "sde_bn_bwd_from_part": ([_P, _I, _P, _P, _P, _LG, _I, _I, _P, _P, _P, _I, _P, _P], c_int),
    "sde_conv_set_halo_min_blocks": ([_I], c_int),
    "sde_conv_set_option": ([_I, _I], c_int),
    "sde_conv_wgrad_splits": ([POINTER(ConvDesc), _I], c_int),
    "sde_conv_wgrad": ([POINTER(ConvDesc), _P, _I, _I, _I, _P, _I, _P, _I, _P], c_int),
    "sde_conv_wgrad_partial": ([POINTER(ConvDesc), _P, _I, _I, _P, _I, _P], c_int),
    "sde_wgrad_reduce_batched": ([_P, _I, _P], c_int),
    "sde_colsum_finalize_batched": ([_P, _I, _P], c_int),
    "sde_prep_input": ([_P, _P, _P, _I, _I, _I, _I, _I, _I, _I, _P, _P], c_int),
    "sde_bn_finalize": ([_P, _I, _I, _LG, _P, _P, _P, _P, _F, _F, _P, _P], c_int),
    "sde_bn_eval_params": ([_P, _P, _P, _P, _F, _I, _P, _P], c_int),
    "sde_bn_apply": ([_P, _P, _P, _I, _LG, _I, _I, _P, _P], c_int),
    "sde_bn_finalize_apply_ok": ([_I, _I, _I], c_int),
    "sde_bn_set_fuse": ([_I], c_int),
    "sde_bn_finalize_apply": ([_P, _I, _I, _LG, _P, _P, _P, _P, _F, _F, _P, _P, _P, _I, _I, _P, _P], c_int),
    "sde_reduce_num_blocks": ([_LG, _I], c_int),
    "sde_bn_bwd": ([_P, _P, _P, _P, _P, _P, _P, _I, _LG, _I, _I, _P, _P, _P, _P, _I, _P, _P, _P], c_int),
    "sde_maxpool_fwd": ([_P, _I, _I, _I, _I, _I, _P, _P, _P], c_int),
    "sde_depth_head_bias_blocks": ([_I, _I, _I], c_int),
    "sde_depth_head_bwd_bias": ([_P, _P, _I, _I, _I, _I, _F, _F, _I, _I, _P, _P, _P, _I, _P], c_int),
    "sde_maxpool_bwd": ([_P, _P, _I, _I, _I, _I, _I, _P, _P], c_int),
    "sde_maxpool_bwd_sum": ([_P, _P, _P, _I, _I, _I, _I, _I, _P, _P], c_int),
    "sde_act_bwd_bias": ([_P, _P, _I, _LG, _I, _I, _P, _P, _P, _I, _I, _P], c_int),
    "sde_act_bwd_bias_sum": ([_P, _P, _P, _I, _LG, _I, _I, _P, _P, _P, _I, _I, _P], c_int),
    "sde_refl_fold": ([_P, _I, _I, _I, _I, _I, _I, _I, _P, _P, _P], c_int),
    "sde_depth_head_fwd": ([_P, _I, _I, _I, _I, _F, _F, _I, _I, _P, _P], c_int),
    "sde_depth_head_bwd": ([_P, _P, _I, _I, _I, _I, _F, _F, _I, _I, _P, _P], c_int),
    "sde_gn_relu_fwd": ([_P, _P, _P, _I, _I, _I, _I, _F, _I, _I, _P, _P, _P, _P], c_int),
    "sde_gn_relu_bwd": ([_P, _P, _P, _P, _P, _I, _I, _I, _I, _I, _I, _P, _P, _P, _P, _I, _P, _P], c_int),
    "sde_gn_relu_res_fwd": ([_P, _P, _P, _P, _I, _I, _I, _I, _F, _I, _I, _P, _P, _P, _P], c_int),
    "sde_gn_relu_res_bwd": ([_P, _P, _P, _P, _P, _P, _I, _I, _I, _I, _I, _I, _P, _P, _P, _P, _I, _P, _P], c_int),
    "sde_space_to_depth": ([_P, _I, _I, _I, _I, _I, _P, _P], c_int),
    "sde_depth_to_space": ([_P, _I, _I, _I, _I, _I, _P, _P], c_int),
    "sde_concat_fwd": ([_P, _P, _P, _I, _I, _I, _I, _I, _I, _I, _I, _P, _P], c_int),
    "sde_concat_bwd": ([_P, _I, _I, _I, _I, _I, _I, _I, _I, _P, _P, _P, _P], c_int),
    "sde_inv_depth_head_fwd": ([_P, _I, _I, _I, _I, _F, _F, _F, _I, _I, _P, _P, _P], c_int),
    "sde_inv_depth_head_bwd": ([_P, _P, _P, _I, _I, _I, _I, _F, _F, _F, _I, _I, _P, _P], c_int),
    "sde_conv3d_fwd": ([_P, _P, _P, _I, _I, _I, _I, _I, _P, _P], c_int),
    "sde_conv3d_dgrad": ([_P, _P, _I, _I, _I, _I, _I, _P, _P], c_int),
    "sde_conv3d_wgrad_num_blocks": ([_I, _I, _I, _I, _I], c_int),
    "sde_conv3d_wgrad": ([_P, _P, _I, _I, _I, _I, _I, _P, _P, _P, _I, _P], c_int),
    "sde_adam_step": ([_P, _P, _P, _P, _LG, _P, _P], c_int),
    "sde_grad_check": ([_P, _LG, _P, _P], c_int),
    "sde_loss_scale_update": ([_P, _F, _F, _I, _P], c_int),
})


OPT_PGEMM, OPT_PGEMM_DEPTH, OPT_PGEMM_3X3, OPT_PGEMM_TILE, OPT_SPLITK, OPT_WGRAD_BLOCKS, OPT_WGRAD_HALO, OPT_CONV_SMALL, OPT_WGRAD_DMA, OPT_BNBWD_FUSE, OPT_CU_RESERVE = 1, 2, 3, 4, 5, 6, 7, 8, 9, 10, 11      # SDE_OPT_* of include/sde_hip.h


def set_option(key, value):
    """sde_conv_set_option: returns the previous value."""
    old = L.lib().sde_conv_set_option(int(key), int(value))
    if old < 0:
        raise L.SdeHipError(f"sde_conv_set_option({key}, {value}) failed: {L.lib().sde_last_error().decode()}")
    return old


def dtype_code(dt):
    if dt == torch.float32:
        return L.F32
    if dt == torch.bfloat16:
        return L.BF16
    if dt == torch.float16:
        return L.F16
    raise L.SdeHipError(f"unsupported activation dtype {dt}")


def vec_of(dt):
    return 4 if dt == torch.float32 else 8


def pad_to(c, v):
    return (c + v - 1) // v * v


def is_ohwi(t):
    """A 4-D conv weight (or its gradient) stored channels-last: memory order [Cout][KH][KW][Cin] under the usual [Cout,Cin,KH,KW] shape -- the
    order of the packed operands and of the weight-gradient slabs (HipTrainer lays its flat buffers out this way)."""
    return t.dim() == 4 and not t.is_contiguous() and t.permute(0, 2, 3, 1).is_contiguous()


def _dense(t):
    return t.is_contiguous() or is_ohwi(t)


def _grad_slot(p):
    """A parameter's pre-allocated fp32 .grad (HipTrainer points it into the flat gradient buffer): kernels then accumulate
    straight into it and the Function returns None for that input, which skips autograd's per-parameter add kernel."""
    g = p.grad if (p is not None and p.is_leaf) else None
    if g is not None and g.dtype == torch.float32 and _dense(g) and g.shape == p.shape:
        return g
    return None


def _f32(t):
    if t.dtype != torch.float32:
        raise L.SdeHipError("parameters must be float32")
    return t.contiguous()


# ---------------------------------------------------------------------------------------------------------------
# raw (non-autograd) helpers
# ---------------------------------------------------------------------------------------------------------------
def pack_weight(w, dtype, cin_pad, cout_pad, for_dgrad=False):
    w = _f32(w)
    Cout, Cin, KH, KW = w.shape
    shape = (cin_pad, KH, KW, cout_pad) if for_dgrad else (cout_pad, KH, KW, cin_pad)
    out = torch.empty(shape, device=w.device, dtype=dtype)
    L.check(L.lib().sde_pack_weight(L.ptr(w), L.ptr(out), dtype_code(dtype), Cout, Cin, KH, KW, cin_pad, cout_pad, int(for_dgrad), L.stream()),
            "sde_pack_weight")
    return out


def _desc(x0, x1, src_mode, KH, KW, stride, pad, reflect, IH, IW, OH, OW):
    d = ConvDesc()
    B, H0, W0, C0 = x0.shape
    d.x0 = x0.data_ptr(); d.x1 = x1.data_ptr() if x1 is not None else 0
    d.dtype = dtype_code(x0.dtype)
    d.C0, d.C1, d.H0, d.W0, d.IH, d.IW = C0, (x1.shape[3] if x1 is not None else 0), H0, W0, IH, IW
    d.src_mode, d.KH, d.KW, d.stride, d.pad, d.reflect = src_mode, KH, KW, stride, pad, int(reflect)
    d.Bn, d.OH, d.OW = B, OH, OW
    return d


_timed = L.timed


def conv_raw(d, x_dtype, w_packed, bias, act, Cout, ldy, want_stats, device, kind="igemm_fwd", flops=0.0):
    y = torch.empty(d.Bn, d.OH, d.OW, ldy, device=device, dtype=x_dtype)
    stats, tiles = None, 0
    lib = L.lib()
    if want_stats:
        tiles = lib.sde_conv_fwd_tiles_m(ctypes.byref(d), ldy)
        stats = torch.empty(tiles + REDUCE_ROWS, Cout, 2, device=device, dtype=torch.float32)
    variant = lib.sde_conv_fwd_variant(ctypes.byref(d), ldy) if L.PROFILE is not None else 0
    meta = None
    if L.PROFILE is not None:
        esz = 4 if x_dtype == torch.float32 else 2
        meta = dict(M=d.Bn * d.OH * d.OW, N=ldy, K=d.KH * d.KW * (d.C0 + d.C1), k=d.KH, s=d.stride, mode=d.src_mode,
                    bytes=esz * (d.Bn * d.H0 * d.W0 * d.C0 + d.Bn * d.IH * d.IW * d.C1 + d.Bn * d.OH * d.OW * ldy))
    ws_bytes = lib.sde_conv_fwd_ws_bytes(ctypes.byref(d), ldy)           # > 0: small-M, long-K layer that runs split-K
    ws = torch.empty(ws_bytes // 4, device=device, dtype=torch.float32) if ws_bytes else None
    _timed(kind, flops, variant, lambda: L.check(lib.sde_conv_fwd_ws(ctypes.byref(d), L.ptr(w_packed), L.ptr(bias), act, L.ptr(y), Cout, ldy, L.ptr(stats),
                                                                     L.ptr(ws), ws_bytes, L.stream()), "sde_conv_fwd_ws"), meta)
    return y, stats


# ---------------------------------------------------------------------------------------------------------------
# Convolution (+bias, +ELU, optional BatchNorm statistics) with every input mode of the path
# ---------------------------------------------------------------------------------------------------------------
class _Conv2d(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x0, x1, weight, bias, stride, pad, reflect, act, upcat, want_stats, owner=None, n_out=1):
        ctx.set_materialize_grads(False)      # the statistics output never carries a gradient: do not launch zero fills for it
        if n_out > 1 and want_stats:
            raise L.SdeHipError("conv2d: output aliases (n_out) belong to the BatchNorm that follows when statistics are requested")
        if not x0.is_contiguous() or (x1 is not None and not x1.is_contiguous()):
            raise L.SdeHipError("conv2d: NHWC inputs must be contiguous")
        dt = x0.dtype
        V = vec_of(dt)
        B, H0, W0, C0 = x0.shape
        Cout, Cin, KH, KW = weight.shape
        C1 = x1.shape[3] if x1 is not None else 0
        if C0 % V or C1 % V:
            raise L.SdeHipError(f"conv2d: channel counts ({C0},{C1}) must be multiples of {V}")
        if Cin > C0 + C1 or (upcat and Cin != C0 + C1):
            raise L.SdeHipError(f"conv2d: weight expects {Cin} input channels, tensors carry {C0}+{C1}")
        IH, IW = (2 * H0, 2 * W0) if upcat else (H0, W0)
        OH = (IH + 2 * pad - KH) // stride + 1
        OW = (IW + 2 * pad - KW) // stride + 1
        ldy = pad_to(Cout, V)
        d = _desc(x0, x1, SRC_UPCAT if upcat else SRC_PLAIN, KH, KW, stride, pad, reflect, IH, IW, OH, OW)
        pre = getattr(owner, "_packed", None) if owner is not None else None
        if pre is not None and pre[0].dtype == dt and pre[0].shape == (ldy, KH, KW, C0 + C1):
            wp, ctx.wd_pre = pre                                   # operands packed once per step by WeightPacker
        else:
            wp, ctx.wd_pre = pack_weight(weight, dt, C0 + C1, ldy), None
            if owner is not None:
                owner._pack_shapes = (dt, C0 + C1, ldy)            # lets WeightPacker build its job table after a first step
        b32 = _f32(bias) if bias is not None else None
        flops = 2.0 * B * OH * OW * Cout * KH * KW * Cin          # algorithmic (real channels)
        # x0 = relu(BatchNorm(y_bn)) with this convolution as its only consumer: the data gradient below can carry BatchNorm's backward reduction
        ent = _BN_OUT.pop(x0.data_ptr(), None) if _BN_OUT else None
        ctx.bn_in = (ent[1], ent[2]) if (ent is not None and ent[0]() is x0 and x1 is None and not upcat and not reflect and stride == 1) else None
        # x0 = relu(BatchNorm(y_bn) + identity) with this convolution and the next block's skip path as its two consumers: the data gradient below can take over
        # that BatchNorm's whole backward reduce pass, once the skip path's gradient has arrived (_RES_GRAD)
        entr = _BN_OUT_RES.get(x0.data_ptr()) if (_BN_OUT_RES and RESBN_FUSED) else None
        ctx.bn_in_res = (entr[1], entr[2]) if (entr is not None and entr[0]() is x0 and x1 is None and not upcat and not reflect and stride == 1 and ctx.bn_in is None) else None
        y, stats = conv_raw(d, dt, wp, b32, act, Cout, ldy, want_stats, x0.device, "igemm_fwd", flops)
        ctx.save_for_backward(x0, x1, weight, y if act != ACT_NONE else None)
        ctx.params = (weight, bias)
        ctx.cfg = (stride, pad, reflect, act, upcat, bias is not None, IH, IW, OH, OW)
        ctx.want_stats = want_stats
        if HEAD_BIAS_FUSED and bias is not None and act == ACT_NONE and Cout == 1 and not want_stats and n_out == 1:
            if len(_HEAD_SLOT) > 64:
                _HEAD_SLOT.clear()
            _HEAD_DONE.clear()                        # (entries live from a head's backward to its convolution's backward only: none survives into a new forward)
            # a disparity head: depth_head's backward can produce this layer's bias gradient on its way.  The weak reference pins the entry to THIS
            # output tensor object (Function.apply hands the same object to the caller), so an address reused by another tensor never matches
            _HEAD_SLOT[y.data_ptr()] = (bias, weakref.ref(y))
        if want_stats:
            ctx.mark_non_differentiable(stats)
            return y, stats
        if n_out > 1:       # one alias per consumer: backward receives their gradients separately and sde_act_bwd_bias_sum adds them on the fly
            return (y,) + tuple(y.view(y.shape) for _ in range(n_out - 1))
        return y

    @staticmethod
    def backward(ctx, *douts):
        x0, x1, weight, y = ctx.saved_tensors
        grads = [douts[0]] if ctx.want_stats else [g for g in douts if g is not None]
        dy = grads[0] if grads else None
        dy1 = None
        if len(grads) > 1:
            dy1 = grads[1].contiguous()
            for g in grads[2:]:
                dy1 = dy1 + g
        stride, pad, reflect, act, upcat, has_bias, IH, IW, OH, OW = ctx.cfg
        dt = x0.dtype
        V = vec_of(dt)
        lib = L.lib()
        dev = x0.device
        B, H0, W0, C0 = x0.shape
        C1 = x1.shape[3] if x1 is not None else 0
        Cout, Cin, KH, KW = weight.shape
        ldy = pad_to(Cout, V)
        if dy is None:
            return (None,) * 12
        off_main = MAIN_STREAM is not None and torch.cuda.current_stream() != MAIN_STREAM
        if off_main and not L.is_aux_stream(torch.cuda.current_stream()):
            # e.g. part of the forward pass ran under torch.cuda.stream(helper): autograd then replays this node's backward on that stream,
            # underneath fork / join events recorded against the trainer's stream (operands could be recycled while a GEMM still reads them)
            raise L.SdeHipError("conv2d backward is running on a different stream than the one the backward phase started on; the "
                                "weight-gradient side-stream bookkeeping supports one main stream (plus the registered auxiliary stream) only")
        # off_main: a layer of the network that runs on the auxiliary stream (PoseNet): both of its GEMMs stay on that stream, in order -- no fork,
        # no group queue; its slabs still join the phase's batched reduction, which the trainer launches after joining the auxiliary stream
        dy = dy.contiguous()
        M = B * OH * OW
        flops = 2.0 * M * Cout * KH * KW * Cin                    # algorithmic FLOPs of each of dgrad / wgrad
        # 1. activation backward + bias gradient
        dbias = None
        dz = dy
        # (the depth head's backward already summed this one-channel layer's bias gradient into its slot: nothing left to do in this step)
        head_did_bias = has_bias and act == ACT_NONE and dy1 is None and _HEAD_DONE.pop(dy.data_ptr(), None) == id(ctx.params[1])
        if has_bias and not head_did_bias and id(ctx.params[1]) in _HEAD_DONE.values():
            # depth_head's backward already accumulated its share of this bias gradient, but the gradient arriving here is not the tensor it
            # returned: the logit has a second consumer and autograd summed the two -- adding the full column sum on top would count the
            # head's share twice (the slot call accumulates)
            for k in [k for k, v in _HEAD_DONE.items() if v == id(ctx.params[1])]:
                del _HEAD_DONE[k]
            raise L.SdeHipError("conv2d backward: the one-channel output feeding depth_head has a second consumer; the fused disparity-head bias "
                                "gradient supports exactly one (set hip.nn.HEAD_BIAS_FUSED = False for such a graph)")
        if (act != ACT_NONE or has_bias or dy1 is not None) and not head_did_bias:
            nblk = lib.sde_reduce_num_blocks(M, ldy)
            part = torch.empty(nblk + REDUCE_ROWS, ldy, device=dev) if has_bias else None
            bslot = _grad_slot(ctx.params[1]) if has_bias else None
            dbias = (bslot if bslot is not None else torch.empty(Cout, device=dev)) if has_bias else None
            dz = torch.empty_like(dy) if (act != ACT_NONE or dy1 is not None) else None
            # the column partials' final sum is nothing the chain waits for: into a gradient slot it rides in the phase's ONE batched finalize (flush)
            later = bslot is not None and WGRAD_DEFER is not None and BIAS_DEFER and L.PROFILE is None and WGRAD_DEFER.accepts_bias(bslot)
            L.check(lib.sde_act_bwd_bias_sum(L.ptr(dy), L.ptr(dy1), L.ptr(y), act, M, ldy, dtype_code(dt), L.ptr(dz), L.ptr(part), None if later else L.ptr(dbias), Cout,
                                             int(bslot is not None), L.stream()), "sde_act_bwd_bias_sum")
            if later:
                WGRAD_DEFER.add_bias(part, nblk, ldy, Cout, bslot)
            if bslot is not None:
                dbias = None
            if dz is None:
                dz = dy
        st = {"dw": None, "forked": False, "side": None, "dx0": None, "dx1": None}
        need_dx = ctx.needs_input_grad[0] or (x1 is not None and ctx.needs_input_grad[1])
        def do_wgrad():
            # 2. weight gradient -- on the side stream when a data gradient follows, so the two independent GEMMs overlap
            dw = None
            forked = False
            side = None
            if ctx.needs_input_grad[2]:
                if not need_dx and WGRAD_DEFER is not None and WGRAD_DEFER.queue and not off_main and EARLY_TAIL:
                    # a layer without a data gradient (the stem) ends the chain: its weight gradient runs on the main stream, so the group still queued goes
                    # to the side stream NOW, underneath it, instead of after it at the phase's flush
                    WGRAD_DEFER.run_queue()
                d = _desc(x0, x1, SRC_UPCAT if upcat else SRC_PLAIN, KH, KW, stride, pad, reflect, IH, IW, OH, OW)
                splits = lib.sde_conv_wgrad_splits(ctypes.byref(d), Cout)
                wslot = _grad_slot(ctx.params[0])
                meta = None
                if L.PROFILE is not None:
                    esz = 4 if dt == torch.float32 else 2
                    meta = dict(M=M, N=Cout, K=KH * KW * (C0 + C1), k=KH, s=stride, mode=int(upcat), splits=splits,
                                bytes=esz * (B * H0 * W0 * C0 + B * IH * IW * C1 + M * ldy) + 8 * splits * Cout * KH * KW * (C0 + C1))
                forked = need_dx and L.SIDE_STREAM and L.PROFILE is None and not off_main
                if forked and L.FORK_MIN_BYTES:      # (A/B aid, default 0: every layer forks)
                    forked = (dz.numel() + x0.numel() + (x1.numel() if x1 is not None else 0)) * dz.element_size() >= L.FORK_MIN_BYTES
                # layers with very large operands (PackNet's full-resolution 64-channel maps: 190 MB each) fork on their own: holding three of
                # them alive for a group pushes the working set out of the Infinity Cache (PackNet-1A: 60.2 vs 58.4 ms/step when grouped)
                op_bytes = (dz.numel() + x0.numel() + (x1.numel() if x1 is not None else 0)) * dz.element_size()
                grouped = forked and L.WGRAD_GROUP > 1 and WGRAD_DEFER is not None and op_bytes <= L.GROUP_MAX_BYTES
                if forked and not grouped and WGRAD_DEFER is not None:
                    WGRAD_DEFER.run_queue()              # keep the side stream in layer order
                cur = torch.cuda.current_stream()
                if grouped:
                    import contextlib
                    wctx = contextlib.nullcontext()      # the launch happens later, inside WGradReducer.run_queue's side-stream context
                elif forked:
                    side = L.side_stream()
                    if WGRAD_DEFER is not None:
                        WGRAD_DEFER.join_pending(keep=L.JOIN_LAG - 1)      # lagging joins of earlier layers
                    side.wait_stream(cur)
                    wctx = torch.cuda.stream(side)
                else:
                    import contextlib
                    wctx = contextlib.nullcontext()
                dw = wslot if wslot is not None else torch.empty(weight.shape, device=dev)      # a fresh gradient tensor is plain OIHW
                wflags = (1 if wslot is not None else 0) | (2 if is_ohwi(dw) else 0)                # SDE_WREDUCE_ACCUMULATE | SDE_WREDUCE_OHWI
                # small slab stacks wait for the phase's one batched reduction; big ones (ResNet-50's 20-40 MB stacks add up to ~1 GB per step)
                # are summed at once on the side stream while they are still in the Infinity Cache and their block can be recycled
                slab_bytes = 4 * splits * Cout * KH * KW * (C0 + C1)
                defer = WGRAD_DEFER if (wslot is not None and WGRAD_DEFER is not None and slab_bytes <= L.DEFER_MAX_BYTES and WGRAD_DEFER.accepts(wslot)) else None
                # one pixel range and a channels-last slot without channel padding: the GEMM's only "slab" IS the gradient row block -- it writes
                # straight into the (zeroed) flat gradient, nothing to reduce.  (A second use of the same weight in the phase accumulates normally.)
                direct = (splits == 1 and wslot is not None and WGRAD_DEFER is not None and WGRAD_DEFER.accepts(wslot) and (C0 + C1) == Cin
                          and (KH * KW == 1 or is_ohwi(wslot)) and wslot.data_ptr() % 16 == 0 and L.PROFILE is None)
                if direct:
                    WGRAD_DEFER._seen.add(wslot.data_ptr())
                    slab, slab_p, defer = wslot, _wptr(wslot), None
                else:
                    slab = torch.empty(splits, Cout, KH * KW * (C0 + C1), device=dev)
                    slab_p = L.ptr(slab)
                if grouped:
                    side_g = L.side_stream(rotate=False)
                    if direct:
                        def launch(d=d):
                            L.check(lib.sde_conv_wgrad_partial(ctypes.byref(d), L.ptr(dz), Cout, ldy, slab_p, 1, L.stream()), "sde_conv_wgrad_partial")
                    elif defer is not None:
                        defer.add(slab, slab.data_ptr(), splits, wslot, Cout, KH * KW, C0 + C1, Cin, wflags)
                        def launch(d=d, slab=slab):
                            L.check(lib.sde_conv_wgrad_partial(ctypes.byref(d), L.ptr(dz), Cout, ldy, L.ptr(slab), splits, L.stream()), "sde_conv_wgrad_partial")
                            slab.record_stream(side_g)
                    else:
                        def launch(d=d, slab=slab, dw=dw):
                            L.check(lib.sde_conv_wgrad(ctypes.byref(d), L.ptr(dz), Cout, ldy, Cin, L.ptr(slab), splits, _wptr(dw), wflags, L.stream()),
                                    "sde_conv_wgrad")
                            slab.record_stream(side_g)
                    WGRAD_DEFER.queue.append((launch, (dz, x0, x1, slab, dw)))
                    WGRAD_DEFER.queue_bytes += op_bytes
                    # a group closes after WGRAD_GROUP layers or once its operands (kept alive until the group's GEMMs ran) exceed the byte budget
                    fg = WGRAD_DEFER.first_group if WGRAD_DEFER.first_group is not None else L.FIRST_GROUP      # (the second phase of a two-phase backward has its own)
                    prefix = str(fg) if fg else ""      # decimal digits = sizes of the first groups of the phase (33: 3 then 3)
                    limit = int(prefix[WGRAD_DEFER.groups_done]) if WGRAD_DEFER.groups_done < len(prefix) else L.WGRAD_GROUP
                    if len(WGRAD_DEFER.queue) >= limit or WGRAD_DEFER.queue_bytes >= L.GROUP_BUDGET_BYTES:
                        WGRAD_DEFER.run_queue()
                    st["dw"], st["forked"], st["side"] = (None if wslot is not None else dw), False, None
                    return
                with wctx:
                    if direct:
                        L.check(lib.sde_conv_wgrad_partial(ctypes.byref(d), L.ptr(dz), Cout, ldy, slab_p, 1, L.stream()), "sde_conv_wgrad_partial")
                    elif defer is not None:
                        _timed("wgrad", flops, 0, lambda: L.check(lib.sde_conv_wgrad_partial(ctypes.byref(d), L.ptr(dz), Cout, ldy, L.ptr(slab), splits, L.stream()),
                                                                  "sde_conv_wgrad_partial"), meta)
                        defer.add(slab, slab.data_ptr(), splits, wslot, Cout, KH * KW, C0 + C1, Cin, wflags)
                    else:
                        if L.PROFILE is None:
                            L.check(lib.sde_conv_wgrad(ctypes.byref(d), L.ptr(dz), Cout, ldy, Cin, L.ptr(slab), splits, _wptr(dw), wflags, L.stream()),
                                    "sde_conv_wgrad")
                        else:       # the same two launches, timed separately (bench.py's roofline pass: GEMM FLOPs against GEMM time)
                            _timed("wgrad", flops, 0, lambda: L.check(lib.sde_conv_wgrad_partial(ctypes.byref(d), L.ptr(dz), Cout, ldy, L.ptr(slab), splits,
                                                                                                 L.stream()), "sde_conv_wgrad_partial"), meta)
                            one = (WReduceItem * 1)(WReduceItem(slab.data_ptr(), dw.data_ptr(), splits, Cout, KH * KW, C0 + C1, Cin, wflags))
                            _timed("wgrad_reduce", 0.0, 0, lambda: L.check(lib.sde_wgrad_reduce_batched(one, 1, L.stream()), "sde_wgrad_reduce_batched"),
                                   dict(jobs=1))
                    if forked and not direct:
                        slab.record_stream(side)
                if wslot is not None:
                    dw = None
            st["dw"], st["forked"], st["side"] = dw, forked, side

        def do_dgrad():
            # 3. data gradient
            dx0 = dx1 = None
            if ctx.needs_input_grad[0] or (x1 is not None and ctx.needs_input_grad[1]):
                Cv = C0 + C1
                wd = ctx.wd_pre if ctx.wd_pre is not None else pack_weight(weight, dt, Cv, ldy, for_dgrad=True)   # [Cv][KH][KW][ldy], taps flipped
                if reflect:
                    dd = _desc(dz, None, SRC_PLAIN, KH, KW, 1, KH - 1, False, OH, OW, IH + 2, IW + 2)
                    dxp, _ = conv_raw(dd, dt, wd, None, ACT_NONE, Cv, Cv, False, dev, "igemm_dgrad", flops)
                    dx0 = torch.empty_like(x0)
                    dx1 = torch.empty_like(x1) if x1 is not None else None
                    L.check(lib.sde_refl_fold(L.ptr(dxp), B, IH, IW, Cv, C0, int(upcat), dtype_code(dt), L.ptr(dx0), L.ptr(dx1), L.stream()), "sde_refl_fold")
                else:
                    if upcat:
                        raise L.SdeHipError("upsample+concat source is only supported with reflection padding (decoder)")
                    if stride == 1:
                        dd = _desc(dz, None, SRC_PLAIN, KH, KW, 1, KH - 1 - pad, False, OH, OW, IH, IW)
                        g_other = _RES_GRAD.pop(x0.data_ptr(), None) if (ctx.bn_in_res is not None and _RES_GRAD) else None
                        if g_other is not None and not (g_other.shape == x0.shape and g_other.dtype == dt and g_other.is_contiguous()):
                            g_other = None
                        rows = lib.sde_conv_dgrad_bnbwd_res_rows(ctypes.byref(dd), Cv, Cv) if g_other is not None else 0
                        if rows > 0:
                            # gm = (this data gradient + the skip path's gradient) * relu'(x0), with the partials of the BatchNorm behind x0: that BatchNorm's
                            # backward finds both under the gradient's address, skips its reduce pass and does not add the skip gradient again
                            y_bn, bnp = ctx.bn_in_res
                            part = torch.empty(rows + REDUCE_ROWS, Cv, 2, device=dev, dtype=torch.float32)
                            dx0 = torch.empty(B, IH, IW, Cv, device=dev, dtype=dt)
                            variant = lib.sde_conv_fwd_variant(ctypes.byref(dd), Cv) if L.PROFILE is not None else 0
                            meta = dict(M=B * IH * IW, N=Cv, K=KH * KW * ldy, k=KH, s=1, mode=0, bytes=2 * (dz.numel() + 4 * dx0.numel())) if L.PROFILE is not None else None
                            _timed("igemm_dgrad", flops, variant, lambda: L.check(lib.sde_conv_dgrad_bnbwd_res(ctypes.byref(dd), L.ptr(wd), L.ptr(dx0), Cv, Cv, L.ptr(y_bn),
                                                                                                           L.ptr(bnp), L.ptr(part), L.ptr(g_other), L.ptr(x0), L.stream()),
                                                                                  "sde_conv_dgrad_bnbwd_res"), meta)
                            if len(_BN_PART) > 16:
                                _BN_PART.clear()
                            _BN_PART[dx0.data_ptr()] = (y_bn.data_ptr(), part, rows, g_other.data_ptr())
                            st["dx0"], st["dx1"] = dx0, None
                            return
                        if g_other is not None:
                            _RES_GRAD[x0.data_ptr()] = g_other                # (no fused form for this layer after all: nothing consumed)
                        rows = lib.sde_conv_dgrad_bnbwd_rows(ctypes.byref(dd), Cv, Cv) if (ctx.bn_in is not None and BNBWD_FUSED) else 0
                        if rows > 0:
                            # the GEMM's epilogue masks the gradient with relu'(bn(y_bn)) and leaves BatchNorm's (sum gm, sum gm * xhat) partials:
                            # _BatchNormAct.backward finds them under the gradient's address and skips its reduce pass
                            y_bn, bnp = ctx.bn_in
                            part = torch.empty(rows + REDUCE_ROWS, Cv, 2, device=dev, dtype=torch.float32)
                            dx0 = torch.empty(B, IH, IW, Cv, device=dev, dtype=dt)
                            variant = lib.sde_conv_fwd_variant(ctypes.byref(dd), Cv) if L.PROFILE is not None else 0
                            meta = dict(M=B * IH * IW, N=Cv, K=KH * KW * ldy, k=KH, s=1, mode=0, bytes=2 * (dz.numel() + 2 * dx0.numel())) if L.PROFILE is not None else None
                            _timed("igemm_dgrad", flops, variant, lambda: L.check(lib.sde_conv_dgrad_bnbwd(ctypes.byref(dd), L.ptr(wd), L.ptr(dx0), Cv, Cv, L.ptr(y_bn),
                                                                                                       L.ptr(bnp), L.ptr(part), L.stream()), "sde_conv_dgrad_bnbwd"), meta)
                            if len(_BN_PART) > 16:
                                _BN_PART.clear()
                            _BN_PART[dx0.data_ptr()] = (y_bn.data_ptr(), part, rows)
                            st["dx0"], st["dx1"] = dx0, None
                            return
                    elif stride == 2:
                        # virtual zero-inserted gradient image: Z[2i, 2j] = dz[i, j]
                        dd = _desc(dz, None, SRC_ZEROINS, KH, KW, 1, KH - 1 - pad, False, 2 * OH - 1, 2 * OW - 1, IH, IW)
                    else:
                        raise L.SdeHipError(f"stride {stride} not supported")
                    dx0, _ = conv_raw(dd, dt, wd, None, ACT_NONE, Cv, Cv, False, dev, "igemm_dgrad", flops)
            st["dx0"], st["dx1"] = dx0, dx1

        do_wgrad()       # first: launching the data-gradient GEMM ahead of the fork measured 10 % slower end to end
        do_dgrad()
        dw, forked, side, dx0, dx1 = st["dw"], st["forked"], st["side"], st["dx0"], st["dx1"]
        if forked:
            if L.JOIN_LAG > 0 and WGRAD_DEFER is not None:
                # lagging join: the main stream goes on with the next layers' backward and joins this layer's weight-gradient GEMM
                # SDE_JOIN_LAG convolutions later (or at the reducer's flush); the operands are kept alive until then
                ev = torch.cuda.Event()
                ev.record(side)
                WGRAD_DEFER.pending.append((ev, (dz, x0, x1)))
            else:
                torch.cuda.current_stream().wait_stream(side)       # join: dz / x0 / x1 stay alive until both GEMMs are done
        return dx0, dx1, dw, dbias, None, None, None, None, None, None, None, None


def _wptr(t):
    """Device pointer of a dense conv-weight-shaped tensor in either memory order (L.ptr insists on torch-contiguous tensors)."""
    if not (t.is_cuda and _dense(t)):
        raise L.SdeHipError("weight gradient buffer must be a dense CUDA tensor (OIHW or channels-last)")
    return c_void_p(t.data_ptr())


EARLY_TAIL = True       # the last queued weight-gradient group is forked when the chain reaches a layer without a data gradient (A/B: False = at the flush)
BIAS_DEFER = True       # bias-gradient column sums of a backward phase in one launch at its end (False: one finalize launch per layer, on the chain)


class ColsumItem(Structure):
    _fields_ = [("part", c_void_p), ("out", c_void_p), ("rows", c_int32), ("ld", c_int32), ("C", c_int32), ("accumulate", c_int32)]


class WReduceItem(Structure):
    _fields_ = [("slab", c_void_p), ("dw", c_void_p), ("rows", c_int32), ("Cout", c_int32), ("KHW", c_int32), ("Cin_pad", c_int32),
                ("Cin_real", c_int32), ("accumulate", c_int32)]


class WGradReducer:
    """Defers the final slab reduction of every convolution's weight gradient to ONE launch per backward phase.

    While installed (`WGRAD_DEFER = reducer`, done by HipTrainer around backward) each _Conv2d.backward runs only the GEMM (+fold) and
    registers its slabs here; flush() sums them all into the flat gradient.  Nothing persists across steps: slabs come from the caching
    allocator (the graph pool under capture) and are released at flush; the item table is a host array that the C side copies into the
    kernel arguments, so there is no device table to upload, keep alive or re-fill after graph capture."""

    def __init__(self):
        self.jobs, self._seen = [], set()
        self.bias_jobs, self._seen_bias = [], set()      # (partial slab, rows, ld, C, gradient slot) of convolutions whose bias-gradient finalize is deferred
        self.forked = False        # some GEMM of this phase still runs on the side stream (late join)
        self.pending = []          # (event behind a layer's side-stream work, its operands) of convolutions whose join is lagging (SDE_JOIN_LAG)
        self.queue = []            # SDE_WGRAD_GROUP > 1: (launch closure, operands) of layers whose weight-gradient GEMM waits for its group's fork
        self.queue_bytes = 0       # operand bytes held by the queued layers
        self.groups_done = 0       # groups forked so far in this backward phase (the first one may be shorter: hip.lib.FIRST_GROUP)
        self.first_group = None    # this phase's first-group digits when they differ from hip.lib.FIRST_GROUP (HipTrainer: second phase of a two-phase backward)

    def run_queue(self):
        """SDE_WGRAD_GROUP > 1: launch the queued weight-gradient GEMMs of the last few layers behind ONE fork of the side stream (one
        cross-stream edge per group in the captured graph instead of one per layer) and leave one lagging join for the whole group."""
        if not self.queue:
            return
        side = L.side_stream()
        self.join_pending(keep=max(0, L.JOIN_LAG - 1))
        side.wait_stream(torch.cuda.current_stream())
        refs = []
        with torch.cuda.stream(side):
            for launch, operands in self.queue:
                launch()
                refs.append(operands)
        ev = torch.cuda.Event()
        ev.record(side)
        self.pending.append((ev, refs))
        self.queue, self.queue_bytes = [], 0
        self.groups_done += 1

    def join_pending(self, keep=0):
        """Make the current stream wait for all but the newest `keep` lagging weight-gradient GEMMs and release their operands."""
        while len(self.pending) > max(0, keep):
            ev, _refs = self.pending.pop(0)
            torch.cuda.current_stream().wait_event(ev)

    def accepts(self, wslot):
        # a weight used twice in one phase (shared modules) must not be accumulated by two blocks of one launch: the second use reduces at once
        return wslot.data_ptr() not in self._seen

    def accepts_bias(self, bslot):
        # (a bias used twice in one phase must not be accumulated by two blocks of one launch)
        return bslot.data_ptr() not in self._seen_bias

    def add_bias(self, part, rows, ld, C, bslot):
        self._seen_bias.add(bslot.data_ptr())
        self.bias_jobs.append((part, rows, ld, C, bslot))

    def flush_bias(self):
        if self.bias_jobs:
            arr = (ColsumItem * len(self.bias_jobs))(*[ColsumItem(p.data_ptr(), b.data_ptr(), r, ld, C, 1) for p, r, ld, C, b in self.bias_jobs])
            L.check(L.lib().sde_colsum_finalize_batched(arr, len(self.bias_jobs), L.stream()), "sde_colsum_finalize_batched")
        self.bias_jobs, self._seen_bias = [], set()

    def add(self, slab, src_ptr, rows, wslot, Cout, KHW, Cin_pad, Cin_real, flags=1):
        self._seen.add(wslot.data_ptr())
        self.jobs.append((slab, wslot, (src_ptr, wslot.data_ptr(), rows, Cout, KHW, Cin_pad, Cin_real, flags)))

    def flush(self):
        self.run_queue()
        self.groups_done = 0
        self.join_pending()
        if self.forked:
            for st_ in L.all_side_streams():
                torch.cuda.current_stream().wait_stream(st_)
            self.forked = False
        self.flush_bias()
        if not self.jobs:
            return
        items = []
        for slab, wslot, it in self.jobs:
            src, dwp, rows, Cout, KHW, Cin_pad, Cin_real, _ = it
            # host-side operand check: the rows to sum must lie inside the slab tensor, the gradient slot must have the OIHW size
            lo, hi = slab.data_ptr(), slab.data_ptr() + slab.numel() * 4
            if not (src is not None and lo <= src and src + rows * Cout * KHW * Cin_pad * 4 <= hi and wslot.numel() == Cout * Cin_real * KHW
                    and dwp == wslot.data_ptr() and Cin_real <= Cin_pad and rows >= 1):
                raise L.SdeHipError(f"WGradReducer: inconsistent job (slab {tuple(slab.shape)}, src offset {None if src is None else src - lo}, "
                                    f"rows {rows}, Cout {Cout}, KHW {KHW}, Cin_pad {Cin_pad}, Cin_real {Cin_real}, slot {tuple(wslot.shape)})")
            items.append(WReduceItem(*it))
        arr = (WReduceItem * len(items))(*items)
        _timed("wgrad_reduce", 0.0, 0, lambda: L.check(L.lib().sde_wgrad_reduce_batched(arr, len(items), L.stream()), "sde_wgrad_reduce_batched"),
               dict(jobs=len(items)))
        self.jobs, self._seen = [], set()


FUSE_BN_FINALIZE = True  # BatchNorm finalize + apply in one launch where the partial slab is short (A/B: set False)
BNBWD_FUSED = True      # BatchNorm's backward reduce pass in the epilogue of the data-gradient GEMM that produces its incoming gradient (A/B, tests: False)
BNBWD_HITS = 0          # times the fused path ran (tests)
RESBN_FUSED = os.environ.get("SDE_RESBN", "1") != "0"      # ... and the whole reduce pass of a residual BatchNorm in the data gradient of the next block's first convolution (A/B, tests: False)
RESBN_HITS = 0
_BN_OUT_RES = {}        # data_ptr of a residual BatchNorm+ReLU output with two consumers -> (weakref to it, y, bnp)
_RES_GRAD = {}          # data_ptr of a block input -> the gradient that arrived over the block's skip path (set by the block's last BatchNorm backward)
_BN_OUT = {}            # data_ptr of a residual-free BatchNorm+ReLU output -> (weakref to it, y, bnp): set by _BatchNormAct.forward, taken by _Conv2d.forward
_BN_PART = {}           # data_ptr of the masked gradient a fused data-gradient GEMM returned -> (y.data_ptr(), partial slab, rows)
WGRAD_DEFER = None      # HipTrainer installs a WGradReducer around backward
MAIN_STREAM = None      # ... and the stream the backward phase runs on: the side-stream fork / lagging-join bookkeeping assumes ONE main stream
FOLD_ROWS = 16          # SDE_WGRAD_FOLD_ROWS


class PackItem(Structure):
    _fields_ = [("src", c_void_p), ("dst_fwd", c_void_p), ("dst_dgrad", c_void_p), ("Cout", c_int32), ("Cin", c_int32), ("KH", c_int32), ("KW", c_int32),
                ("Cin_pad", c_int32), ("Cout_pad", c_int32), ("src_layout", c_int32), ("reserved", c_int32), ("end", ctypes.c_int64)]


class WeightPacker:
    """Packs the forward and data-gradient operands of every convolution of a model in ONE kernel launch per step.

    Build it after one forward pass (each HipConv2d then knows the padded shapes its operands need); call run() whenever the master
    weights changed (HipTrainer does, at the start of every step, inside the captured graph)."""

    def __init__(self, model):
        import numpy as np
        convs = [m for m in model.modules() if getattr(m, "_pack_shapes", None) is not None]
        if not convs:
            raise L.SdeHipError("WeightPacker: run one forward pass first")
        self.dtype = convs[0]._pack_shapes[0]
        items, end = [], 0
        self._keep = []
        for m in convs:
            dt, cin_pad, ldy = m._pack_shapes
            if dt != self.dtype:
                raise L.SdeHipError("WeightPacker: mixed compute dtypes")
            w = m.weight
            Cout, Cin, KH, KW = w.shape
            wp = torch.empty(ldy, KH, KW, cin_pad, device=w.device, dtype=dt)
            wd = torch.empty(cin_pad, KH, KW, ldy, device=w.device, dtype=dt)
            m._packed = (wp, wd)
            nb = L.lib().sde_pack_item_blocks(ldy, cin_pad, KH, KW)
            if nb <= 0:
                raise L.SdeHipError(f"WeightPacker: unsupported kernel size {KH}x{KW}")
            end += nb
            if not _dense(w):
                raise L.SdeHipError("WeightPacker: conv weights must be dense (OIHW or channels-last)")
            items.append((w.data_ptr(), wp.data_ptr(), wd.data_ptr(), Cout, Cin, KH, KW, cin_pad, ldy, 1 if is_ohwi(w) else 0, 0, end))
            self._keep.append((w, wp, wd))
        dev = convs[0].weight.device

        def table(rows):
            arr = (PackItem * len(rows))(*[PackItem(*it) for it in rows])
            return torch.from_numpy(np.frombuffer(bytes(arr), dtype=np.uint8).copy()).to(dev)
        self.items = table(items)
        self.n, self.total = len(items), end
        self._ptrs = [w.data_ptr() for w, _, _ in self._keep]

    def run(self):
        if any(w.data_ptr() != p for (w, _, _), p in zip(self._keep, self._ptrs)):
            raise L.SdeHipError("WeightPacker: a weight tensor moved (e.g. flattened after the packer was built); rebuild the packer")
        L.check(L.lib().sde_pack_weights_batched(L.ptr(self.items), self.n, self.total, dtype_code(self.dtype), L.stream()), "sde_pack_weights_batched")

    def join_dgrad(self):
        """Kept for the trainer's call order (the data-gradient operands are packed by the same launch as the forward ones)."""


def conv2d(x, weight, bias=None, stride=1, pad=0, reflect=False, act=ACT_NONE, skip=None, upsample=False, bn_stats=False, owner=None, n_out=1):
    """y = act(conv(x) + bias) on NHWC tensors.

    upsample=True: the input is cat(nearest_x2(x), skip) (skip may be None) -- depth_decoder.py:L102-105 -- gathered on the fly.
    bn_stats=True additionally returns the per-tile (sum, sum^2) slab BatchNorm needs.
    n_out > 1 returns that many aliases of the output, one per consumer: their gradients are summed by the activation-backward kernel instead
    of by an autograd add kernel per extra consumer.
    """
    return _Conv2d.apply(x, skip, weight, bias, int(stride), int(pad), bool(reflect), int(act), bool(upsample), bool(bn_stats), owner, int(n_out))


# ---------------------------------------------------------------------------------------------------------------
# BatchNorm (+ReLU, +residual)
# ---------------------------------------------------------------------------------------------------------------
class _BatchNormAct(torch.autograd.Function):
    """BatchNorm (+residual, +ReLU).  n_out > 1 returns that many aliases of the output, one per consumer (next block's first convolution, its
    residual / down-sampling path, a decoder skip): backward then receives the consumers' gradients separately and the kernels sum them on the
    fly, instead of autograd launching an add kernel per extra consumer (23 per step on ResNet-50, 1.5 GB of traffic)."""

    @staticmethod
    def forward(ctx, y, stats, gamma, beta, running_mean, running_var, residual, relu, momentum, eps, training, n_out):
        ctx.set_materialize_grads(False)
        dt = y.dtype
        C = y.shape[-1]
        M = y.numel() // C
        lib = L.lib()
        dev = y.device
        bnp = torch.empty(4, C, device=dev)
        out = torch.empty_like(y)
        tiles = stats.shape[0] - REDUCE_ROWS if training else 0
        if training and FUSE_BN_FINALIZE and lib.sde_bn_finalize_apply_ok(tiles, C, dtype_code(dt)):
            # short partial slabs (one row per persistent workgroup of the producing GEMM): finalize + apply in one launch
            L.check(lib.sde_bn_finalize_apply(L.ptr(stats), tiles, C, M, L.ptr(_f32(gamma)), L.ptr(_f32(beta)), L.ptr(running_mean), L.ptr(running_var),
                                              momentum, eps, L.ptr(bnp), L.ptr(y), L.ptr(residual), int(relu), dtype_code(dt), L.ptr(out), L.stream()),
                    "sde_bn_finalize_apply")
        else:
            if training:
                L.check(lib.sde_bn_finalize(L.ptr(stats), tiles, C, M, L.ptr(_f32(gamma)), L.ptr(_f32(beta)), L.ptr(running_mean), L.ptr(running_var),
                                            momentum, eps, L.ptr(bnp), L.stream()), "sde_bn_finalize")
            else:
                L.check(lib.sde_bn_eval_params(L.ptr(_f32(gamma)), L.ptr(_f32(beta)), L.ptr(running_mean), L.ptr(running_var), eps, C, L.ptr(bnp), L.stream()),
                        "sde_bn_eval_params")
            L.check(lib.sde_bn_apply(L.ptr(y), L.ptr(bnp), L.ptr(residual), int(relu), M, C, dtype_code(dt), L.ptr(out), L.stream()), "sde_bn_apply")
        # backward needs the ReLU mask: with a residual it comes from the saved output; without one sde_bn_bwd re-derives it from y and the
        # BatchNorm parameters (out = NULL), so that tensor is neither kept for backward nor read by it
        ctx.save_for_backward(y, out if (relu and residual is not None) else None, bnp, gamma)
        ctx.params = (gamma, beta)
        ctx.cfg = (relu, residual is not None, training)
        if training and relu and residual is None and n_out == 1 and dt != torch.float32 and C % 64 == 0 and ctx.needs_input_grad[0] and BNBWD_FUSED:
            # candidates for the fused backward reduction: the convolution that consumes `out` (and nothing else does: n_out == 1) picks this up
            if len(_BN_OUT) > 64:
                _BN_OUT.clear()
            _BN_PART.clear()                          # (entries live from a convolution's backward to this BatchNorm's backward only)
            _BN_OUT[out.data_ptr()] = (weakref.ref(out), y, bnp)
        ctx.res_ptr = residual.data_ptr() if residual is not None else None
        if training and relu and residual is not None and n_out == 2 and dt != torch.float32 and C % 64 == 0 and ctx.needs_input_grad[0] and BNBWD_FUSED and RESBN_FUSED:
            # residual BatchNorm + ReLU with two consumers (torchvision's blocks: the next block's first convolution and its skip path): that convolution's
            # data gradient can carry this BatchNorm's backward reduce pass (residual form), see _Conv2d.backward
            if len(_BN_OUT_RES) > 64:
                _BN_OUT_RES.clear()
            _RES_GRAD.clear()                         # (entries live from a block's last BatchNorm backward to its first convolution's backward only)
            _BN_OUT_RES[out.data_ptr()] = (weakref.ref(out), y, bnp)
            ctx.res_tag = out.data_ptr()              # (this BatchNorm's backward drops the entry: the consuming convolution took its references in forward)
        if n_out == 1:
            return out
        return (out,) + tuple(out.view(out.shape) for _ in range(n_out - 1))

    @staticmethod
    def backward(ctx, *douts):
        y, out, bnp, gamma = ctx.saved_tensors
        relu, has_res, training = ctx.cfg
        if getattr(ctx, "res_tag", None) is not None:
            _BN_OUT_RES.pop(ctx.res_tag, None)
        if not training:
            raise L.SdeHipError("BatchNorm backward in eval mode is not on the path")
        grads = [d.contiguous() for d in douts if d is not None]
        if not grads:
            return (None,) * 12
        if len(grads) > 3:
            extra = grads[3]
            for g in grads[4:]:
                extra = extra + g
            grads = grads[:2] + [grads[2] + extra]
        dt = y.dtype
        C = y.shape[-1]
        M = y.numel() // C
        lib = L.lib()
        dev = y.device
        coef = torch.empty(2, C, device=dev)
        gs, bs = _grad_slot(ctx.params[0]), _grad_slot(ctx.params[1])
        direct = gs is not None and bs is not None
        dgamma = gs if direct else torch.empty(C, device=dev)
        dbeta = bs if direct else torch.empty(C, device=dev)
        dy = torch.empty_like(y)
        ent = _BN_PART.pop(grads[0].data_ptr(), None) if (_BN_PART and len(grads) == 1 and relu and not has_res) else None
        if ent is not None and ent[0] == y.data_ptr() and grads[0].shape == y.shape and grads[0].dtype == dt:
            # the data-gradient GEMM that produced this gradient already masked it and reduced it (sde_conv_dgrad_bnbwd): finalize + apply only
            global BNBWD_HITS
            BNBWD_HITS += 1
            L.check(lib.sde_bn_bwd_from_part(L.ptr(ent[1]), ent[2], L.ptr(grads[0]), L.ptr(y), L.ptr(bnp), M, C, dtype_code(dt), L.ptr(coef), L.ptr(dgamma),
                                             L.ptr(dbeta), int(direct), L.ptr(dy), L.stream()), "sde_bn_bwd_from_part")
            if direct:
                dgamma = dbeta = None
            return dy, None, dgamma, dbeta, None, None, None, None, None, None, None, None
        entr = _BN_PART.get(grads[0].data_ptr()) if (_BN_PART and len(grads) == 2 and relu and has_res) else None
        if entr is not None and len(entr) == 4:
            # residual form: the next block's first convolution already formed gm = (its data gradient + the skip gradient) * relu'(out) and reduced it
            del _BN_PART[grads[0].data_ptr()]
            if not (entr[0] == y.data_ptr() and entr[3] == grads[1].data_ptr() and grads[0].shape == y.shape and grads[0].dtype == dt):
                raise L.SdeHipError("BatchNorm backward: a data gradient that already contains the skip path's gradient arrived at a BatchNorm it was not "
                                    "formed for (hip.nn.RESBN_FUSED = False selects the separate reduce pass)")
            global RESBN_HITS
            RESBN_HITS += 1
            gm = grads[0]
            L.check(lib.sde_bn_bwd_from_part(L.ptr(entr[1]), entr[2], L.ptr(gm), L.ptr(y), L.ptr(bnp), M, C, dtype_code(dt), L.ptr(coef), L.ptr(dgamma),
                                             L.ptr(dbeta), int(direct), L.ptr(dy), L.stream()), "sde_bn_bwd_from_part")
            if ctx.res_ptr is not None:
                _RES_GRAD[ctx.res_ptr] = gm
            if direct:
                dgamma = dbeta = None
            return dy, None, dgamma, dbeta, None, None, gm, None, None, None, None, None
        part = torch.empty(lib.sde_reduce_num_blocks(M, C) + REDUCE_ROWS, C, 2, device=dev)
        # gm = relu'(out) * (sum of the incoming gradients): needed when there is anything to mask or to sum; it is also the residual's gradient
        gm = torch.empty_like(y) if (relu or len(grads) > 1) else None
        d0, d1, d2 = (grads + [None, None])[:3]
        L.check(lib.sde_bn_bwd(L.ptr(d0), L.ptr(d1), L.ptr(d2), L.ptr(out), L.ptr(y), L.ptr(bnp), L.ptr(gamma), int(relu), M, C, dtype_code(dt), L.ptr(part),
                               L.ptr(coef), L.ptr(dgamma), L.ptr(dbeta), int(direct), L.ptr(gm), L.ptr(dy), L.stream()), "sde_bn_bwd")
        dres = (gm if gm is not None else d0) if has_res else None
        if dres is not None and ctx.res_ptr is not None and RESBN_FUSED and _BN_OUT_RES:
            _RES_GRAD[ctx.res_ptr] = dres          # the skip path's gradient, for the data gradient of the convolution that shares the block input (residual form)
        if direct:
            dgamma = dbeta = None
        return dy, None, dgamma, dbeta, None, None, dres, None, None, None, None, None


def batch_norm_act(y, stats, gamma, beta, running_mean, running_var, residual=None, relu=True, momentum=0.1, eps=1e-5, training=True, n_out=1):
    return _BatchNormAct.apply(y, stats, gamma, beta, running_mean, running_var, residual, bool(relu), float(momentum), float(eps), bool(training), int(n_out))


# ---------------------------------------------------------------------------------------------------------------
# MaxPool 3x3 / 2
# ---------------------------------------------------------------------------------------------------------------
class _MaxPool(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x, n_out=1):
        ctx.set_materialize_grads(False)
        B, H, W, C = x.shape
        OH, OW = (H - 1) // 2 + 1, (W - 1) // 2 + 1
        out = torch.empty(B, OH, OW, C, device=x.device, dtype=x.dtype)
        idx = torch.empty(B, OH, OW, C, device=x.device, dtype=torch.uint8)
        L.check(L.lib().sde_maxpool_fwd(L.ptr(x.contiguous()), B, H, W, C, dtype_code(x.dtype), L.ptr(out), L.ptr(idx), L.stream()), "sde_maxpool_fwd")
        ctx.save_for_backward(idx)
        ctx.shape = (B, H, W, C)
        if n_out > 1:
            return (out,) + tuple(out.view(out.shape) for _ in range(n_out - 1))
        return out

    @staticmethod
    def backward(ctx, *douts):
        (idx,) = ctx.saved_tensors
        B, H, W, C = ctx.shape
        grads = [g.contiguous() for g in douts if g is not None]
        if not grads:
            return None, None
        d0, d1 = grads[0], None
        if len(grads) > 1:
            d1 = grads[1]
            for g in grads[2:]:
                d1 = d1 + g
        dx = torch.empty(B, H, W, C, device=d0.device, dtype=d0.dtype)
        L.check(L.lib().sde_maxpool_bwd_sum(L.ptr(d0), L.ptr(d1), L.ptr(idx), B, H, W, C, dtype_code(d0.dtype), L.ptr(dx), L.stream()), "sde_maxpool_bwd_sum")
        return dx, None


def max_pool_3x3_s2(x, n_out=1):
    """n_out > 1: that many aliases of the pooled tensor (layer1's first block reads it twice); backward sums their gradients in the kernel."""
    return _MaxPool.apply(x, int(n_out))


# ---------------------------------------------------------------------------------------------------------------
# Input preparation and the depth head tail
# ---------------------------------------------------------------------------------------------------------------
def prep_input(img, mean, std, dtype, flip=False):
    """NCHW fp32 image -> (img - mean)/std as NHWC `dtype` with channels padded to 16 bytes (no autograd: images carry no grad)."""
    if img.dtype != torch.float32:
        raise L.SdeHipError("prep_input expects a float32 NCHW image")
    img = img.contiguous()
    B, C, H, W = img.shape
    Cp = pad_to(C, vec_of(dtype))
    out = torch.empty(B, H, W, Cp, device=img.device, dtype=dtype)
    m = _f32(mean.reshape(-1)) if mean is not None else None
    s = _f32(std.reshape(-1)) if std is not None else None
    L.check(L.lib().sde_prep_input(L.ptr(img), L.ptr(m), L.ptr(s), B, C, H, W, Cp, int(bool(flip)), dtype_code(dtype), L.ptr(out), L.stream()), "sde_prep_input")
    return out


HEAD_BIAS_HITS = 0       # times the fused path below ran (tests)
HEAD_BIAS_FUSED = True   # the bias gradient of a one-channel convolution feeding depth_head comes out of depth_head's backward (False: separate pass; tests)
_HEAD_SLOT = {}     # data_ptr of a one-channel bias convolution's output -> its bias parameter (set by _Conv2d.forward, taken by _DepthHead.forward)
_HEAD_DONE = {}     # data_ptr of the logit gradient _DepthHead.backward returned -> id(bias parameter) whose gradient it accumulated


class _DepthHead(torch.autograd.Function):
    @staticmethod
    def forward(ctx, y, min_depth, max_depth, flip):
        ent = _HEAD_SLOT.pop(y.data_ptr(), None)
        ctx.bias_param = ent[0] if (ent is not None and ent[1]() is y) else None
        B, H, W, ld = y.shape
        depth = torch.empty(B, 1, H, W, device=y.device, dtype=torch.float32)
        L.check(L.lib().sde_depth_head_fwd(L.ptr(y.contiguous()), B, H, W, ld, min_depth, max_depth, int(flip), dtype_code(y.dtype), L.ptr(depth), L.stream()),
                "sde_depth_head_fwd")
        ctx.save_for_backward(y)
        ctx.cfg = (min_depth, max_depth, flip)
        return depth

    @staticmethod
    def backward(ctx, ddepth):
        (y,) = ctx.saved_tensors
        min_depth, max_depth, flip = ctx.cfg
        B, H, W, ld = y.shape
        dy = torch.empty_like(y)
        lib = L.lib()
        bslot = _grad_slot(ctx.bias_param) if ctx.bias_param is not None else None
        if bslot is not None and bslot.numel() == 1:
            # the convolution in front has ONE output channel: its bias gradient is the sum of the logit gradients this kernel writes -- summed here,
            # into the flat gradient, instead of by a separate pass over the (8-channel padded) gradient tensor in the convolution's backward
            part = torch.empty(lib.sde_depth_head_bias_blocks(B, H, W), device=y.device)
            L.check(lib.sde_depth_head_bwd_bias(L.ptr(y), L.ptr(ddepth.contiguous().float()), B, H, W, ld, min_depth, max_depth, int(flip), dtype_code(y.dtype),
                                                L.ptr(dy), L.ptr(part), L.ptr(bslot), 1, L.stream()), "sde_depth_head_bwd_bias")
            global HEAD_BIAS_HITS
            HEAD_BIAS_HITS += 1
            _HEAD_DONE[dy.data_ptr()] = id(ctx.bias_param)
        else:
            L.check(lib.sde_depth_head_bwd(L.ptr(y), L.ptr(ddepth.contiguous().float()), B, H, W, ld, min_depth, max_depth, int(flip), dtype_code(y.dtype),
                                           L.ptr(dy), L.stream()), "sde_depth_head_bwd")
        return dy, None, None, None


def depth_head(y, min_depth, max_depth, flip=False):
    """softplus + disp_to_depth(...)[1] (+ flip) on channel 0 of y -> [B,1,H,W] fp32 depth."""
    return _DepthHead.apply(y, float(min_depth), float(max_depth), bool(flip))


# ---------------------------------------------------------------------------------------------------------------
# GroupNorm + ReLU
# ---------------------------------------------------------------------------------------------------------------
class _GroupNormReLU(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x, gamma, beta, groups, eps, relu, res=None):
        B, H, W, C = x.shape
        dev = x.device
        part = torch.empty(B, GN_CHUNKS, C, 2, device=dev)
        gnp = torch.empty(B, groups, 2, device=dev)
        out = torch.empty_like(x)
        x = x.contiguous()
        if res is not None:
            if res.shape != x.shape or res.dtype != x.dtype:
                raise L.SdeHipError("group_norm_relu: the residual must have the input's shape and dtype")
            res = res.contiguous()
        L.check(L.lib().sde_gn_relu_res_fwd(L.ptr(x), L.ptr(res), L.ptr(_f32(gamma)), L.ptr(_f32(beta)), B, H * W, C, groups, eps, int(relu), dtype_code(x.dtype),
                                            L.ptr(part), L.ptr(gnp), L.ptr(out), L.stream()), "sde_gn_relu_res_fwd")
        ctx.save_for_backward(x, out, gnp, gamma, res)
        ctx.params = (gamma, beta)
        ctx.cfg = (groups, relu)
        return out

    @staticmethod
    def backward(ctx, dout):
        x, out, gnp, gamma, res = ctx.saved_tensors
        groups, relu = ctx.cfg
        B, H, W, C = x.shape
        dev = x.device
        part = torch.empty(B, GN_CHUNKS, C, 2, device=dev)
        coef = torch.empty(B, groups, 2, device=dev)
        gs, bs = _grad_slot(ctx.params[0]), _grad_slot(ctx.params[1])
        direct = gs is not None and bs is not None
        dgamma = gs if direct else torch.empty(C, device=dev)
        dbeta = bs if direct else torch.empty(C, device=dev)
        dx = torch.empty_like(x)
        L.check(L.lib().sde_gn_relu_res_bwd(L.ptr(dout.contiguous()), L.ptr(out), L.ptr(x), L.ptr(res), L.ptr(gnp), L.ptr(_f32(gamma)), B, H * W, C, groups, int(relu),
                                            dtype_code(x.dtype), L.ptr(part), L.ptr(coef), L.ptr(dgamma), L.ptr(dbeta), int(direct), L.ptr(dx), L.stream()),
                "sde_gn_relu_res_bwd")
        if direct:
            dgamma = dbeta = None
        # the normalised tensor is x + res: both receive the same gradient (one tensor, no copy)
        return dx, dgamma, dbeta, None, None, None, (dx if res is not None else None)


GN_CHUNKS = 64          # SDE_GN_CHUNKS
GN_ACT = {False: 0, True: 1, "none": 0, "relu": 1, "elu": 2}


def group_norm_relu(x, gamma, beta, groups=16, eps=1e-5, relu=True, residual=None):
    """GroupNorm + activation; relu: True / "relu" (PoseNet.py:L13-20), "elu" (layers01.py:L33-40), False / "none".
    residual: the normalised tensor is x + residual (layers01.py:L74-76), summed inside the kernels."""
    if x.shape[-1] != gamma.numel():
        raise L.SdeHipError("group_norm_relu: channel padding is not supported (PoseNet / PackNet channels are multiples of 16)")
    return _GroupNormReLU.apply(x, gamma, beta, int(groups), float(eps), GN_ACT[relu], residual)


# ---------------------------------------------------------------------------------------------------------------
# PackNet01's data movement (csrc/packnet.hip): space-to-depth / depth-to-space, channel concatenation, the inverse-depth head
# ---------------------------------------------------------------------------------------------------------------
def _s2d(x, inverse):
    x = x.contiguous()
    B, H, W, C = x.shape
    if inverse:
        y = torch.empty(B, 2 * H, 2 * W, C // 4, device=x.device, dtype=x.dtype)
        L.check(L.lib().sde_depth_to_space(L.ptr(x), B, H, W, C, dtype_code(x.dtype), L.ptr(y), L.stream()), "sde_depth_to_space")
    else:
        y = torch.empty(B, H // 2, W // 2, 4 * C, device=x.device, dtype=x.dtype)
        L.check(L.lib().sde_space_to_depth(L.ptr(x), B, H, W, C, dtype_code(x.dtype), L.ptr(y), L.stream()), "sde_space_to_depth")
    return y


class _SpaceToDepth(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x, inverse):
        ctx.inverse = inverse
        return _s2d(x, inverse)

    @staticmethod
    def backward(ctx, dy):
        return _s2d(dy, not ctx.inverse), None


def space_to_depth(x):
    """layers01.py:L138-160 `packing` (r = 2) on NHWC: out[b,y,x,c*4+dy*2+dx] = in[b,2y+dy,2x+dx,c]."""
    return _SpaceToDepth.apply(x, False)


def depth_to_space(x):
    """nn.PixelShuffle(2) on NHWC: out[b,2y+dy,2x+dx,c] = in[b,y,x,c*4+dy*2+dx]."""
    return _SpaceToDepth.apply(x, True)


class _Concat(torch.autograd.Function):
    @staticmethod
    def forward(ctx, p0, p1, inv, add):
        p0 = p0.contiguous()
        B, H, W, C0 = p0.shape
        V = vec_of(p0.dtype)
        C1 = p1.shape[3] if p1 is not None else 0
        if p1 is not None:
            p1 = p1.contiguous()
            if p1.dtype != p0.dtype or p1.shape[:3] != p0.shape[:3]:
                raise L.SdeHipError("concat: sources differ in dtype or spatial size")
        if inv is not None:
            inv = inv.contiguous()
            if inv.dtype != torch.float32 or tuple(inv.shape) != (B, H // 2, W // 2):
                raise L.SdeHipError(f"concat: the inverse-depth map must be fp32 [B, H/2, W/2], got {tuple(inv.shape)} {inv.dtype}")
        Ct = pad_to(C0 + (0 if add else C1) + (1 if inv is not None else 0), V)
        out = torch.empty(B, H, W, Ct, device=p0.device, dtype=p0.dtype)
        L.check(L.lib().sde_concat_fwd(L.ptr(p0), L.ptr(p1), L.ptr(inv), int(add), B, H, W, C0, C1, Ct, dtype_code(p0.dtype), L.ptr(out), L.stream()), "sde_concat_fwd")
        ctx.cfg = (B, H, W, C0, C1, Ct, add, inv is not None, p1 is not None)
        return out

    @staticmethod
    def backward(ctx, dout):
        B, H, W, C0, C1, Ct, add, has_inv, has_p1 = ctx.cfg
        dout = dout.contiguous()
        d0 = torch.empty(B, H, W, C0, device=dout.device, dtype=dout.dtype)
        d1 = torch.empty(B, H, W, C1, device=dout.device, dtype=dout.dtype) if (has_p1 and not add) else None
        dinv = torch.empty(B, H // 2, W // 2, device=dout.device, dtype=torch.float32) if (has_inv and ctx.needs_input_grad[2]) else None
        L.check(L.lib().sde_concat_bwd(L.ptr(dout), int(add), B, H, W, C0, C1, Ct, dtype_code(dout.dtype), L.ptr(d0), L.ptr(d1), L.ptr(dinv), L.stream()), "sde_concat_bwd")
        return d0, (d0 if (add and has_p1) else d1), dinv, None


def concat(p0, p1=None, inv_depth=None, add=False):
    """cat([p0, p1(, nearest_x2(inv_depth))], channel) -- or [p0 + p1(, ...)] with add=True -- zero-filled to the 16-byte group, in one pass
    (PackNet01.py:L150-199).  inv_depth: fp32 [B, H/2, W/2]."""
    return _Concat.apply(p0, p1, inv_depth, bool(add))


class _InvDepthHead(torch.autograd.Function):
    @staticmethod
    def forward(ctx, y, min_depth_head, min_depth, max_depth, flip):
        y = y.contiguous()
        B, H, W, ld = y.shape
        inv = torch.empty(B, H, W, device=y.device, dtype=torch.float32)
        depth = torch.empty(B, 1, H, W, device=y.device, dtype=torch.float32)
        L.check(L.lib().sde_inv_depth_head_fwd(L.ptr(y), B, H, W, ld, min_depth_head, min_depth, max_depth, int(flip), dtype_code(y.dtype), L.ptr(inv), L.ptr(depth),
                                               L.stream()), "sde_inv_depth_head_fwd")
        ctx.save_for_backward(y)
        ctx.cfg = (min_depth_head, min_depth, max_depth, flip)
        ctx.set_materialize_grads(False)
        return inv, depth

    @staticmethod
    def backward(ctx, d_inv, d_depth):
        (y,) = ctx.saved_tensors
        if d_inv is None and d_depth is None:
            return None, None, None, None, None
        mdh, mn, mx, flip = ctx.cfg
        B, H, W, ld = y.shape
        dy = torch.empty_like(y)
        di = d_inv.contiguous().float() if d_inv is not None else None
        dd = d_depth.contiguous().float() if d_depth is not None else None
        L.check(L.lib().sde_inv_depth_head_bwd(L.ptr(y), L.ptr(di), L.ptr(dd), B, H, W, ld, mdh, mn, mx, int(flip), dtype_code(y.dtype), L.ptr(dy), L.stream()),
                "sde_inv_depth_head_bwd")
        return dy, None, None, None, None


def inv_depth_head(y, min_depth_head, min_depth, max_depth, flip=False):
    """(sigmoid(y[...,0]) / min_depth_head  [B,H,W] fp32,  disp_to_depth(.)[1] [B,1,H,W] fp32, mirrored along x when flip) -- layers01.py:L105-133 and
    PackNet01.py:L120-123,L199."""
    return _InvDepthHead.apply(y, float(min_depth_head), float(min_depth), float(max_depth), bool(flip))


# ---------------------------------------------------------------------------------------------------------------
# PackNet's Conv3d(1, 8, 3, padding=1) over the (channel, y, x) volume (layers01.py:L223-298)
# ---------------------------------------------------------------------------------------------------------------
class _Conv3dPack(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x, weight, bias):
        B, H, W, D = x.shape
        lib = L.lib()
        x = x.contiguous()
        y = torch.empty(B, H, W, 8 * D, device=x.device, dtype=x.dtype)
        L.check(lib.sde_conv3d_fwd(L.ptr(x), L.ptr(_f32(weight)), L.ptr(_f32(bias)), B, H, W, D, dtype_code(x.dtype), L.ptr(y), L.stream()), "sde_conv3d_fwd")
        ctx.save_for_backward(x, weight)
        ctx.params = (weight, bias)
        ctx.set_materialize_grads(False)
        return y

    @staticmethod
    def backward(ctx, dy):
        if dy is None:
            return None, None, None
        x, weight = ctx.saved_tensors
        B, H, W, D = x.shape
        lib = L.lib()
        dy = dy.contiguous()
        dc = dtype_code(x.dtype)
        ws, bs = _grad_slot(ctx.params[0]), _grad_slot(ctx.params[1])
        direct = ws is not None and bs is not None
        dw = ws if direct else torch.empty_like(weight)
        db = bs if direct else torch.empty(8, device=x.device)
        part = torch.empty(lib.sde_conv3d_wgrad_num_blocks(B, H, W, D, dc), 224, device=x.device)
        L.check(lib.sde_conv3d_wgrad(L.ptr(x), L.ptr(dy), B, H, W, D, dc, L.ptr(part), L.ptr(dw), L.ptr(db), int(direct), L.stream()), "sde_conv3d_wgrad")
        dx = None
        if ctx.needs_input_grad[0]:
            dx = torch.empty_like(x)
            L.check(lib.sde_conv3d_dgrad(L.ptr(dy), L.ptr(_f32(weight)), B, H, W, D, dc, L.ptr(dx), L.stream()), "sde_conv3d_dgrad")
        return dx, (None if direct else dw), (None if direct else db)


def conv3d_pack(x, weight, bias):
    """x: NHWC [B,H,W,D]; weight [8,1,3,3,3] fp32; bias [8] -> [B,H,W,8*D] (channel = feature*D + ch, as view(b, c*d, h, w))."""
    if tuple(weight.shape) != (8, 1, 3, 3, 3):
        raise L.SdeHipError(f"conv3d_pack: weight shape {tuple(weight.shape)} (expected [8,1,3,3,3])")
    return _Conv3dPack.apply(x, weight, bias)


# ---------------------------------------------------------------------------------------------------------------
# Fused Adam / AdamW over a flat buffer
# ---------------------------------------------------------------------------------------------------------------
ADAM_MAX_SEG = 8       # SDE_ADAM_MAX_SEG


class AdamDesc(Structure):
    _fields_ = [("seg_end", c_long * ADAM_MAX_SEG), ("seg_lr", c_float * ADAM_MAX_SEG), ("seg_wd", c_float * ADAM_MAX_SEG), ("nseg", c_int32),
                ("decoupled_wd", c_int32), ("beta1", c_float), ("beta2", c_float), ("eps", c_float), ("bias_corr1", c_float), ("bias_corr2", c_float),
                ("grad_scale", c_float), ("scale_state", c_void_p), ("beta1_d", ctypes.c_double), ("beta2_d", ctypes.c_double)]


def adam_step(p, g, m, v, seg_end, seg_lr, seg_wd, bias_corr, beta1=0.9, beta2=0.999, eps=1e-8, grad_scale=1.0, decoupled_wd=False, scale_state=None):
    """seg_end / seg_lr / seg_wd: HOST sequences (one entry per segment); everything travels by value in the kernel arguments.
    scale_state: optional device float[4] {loss_scale, found_inf, growth_tracker, applied_steps} (fp16 dynamic loss scaling); with it the
    kernel forms the bias corrections itself from the device-side count of APPLIED steps and `bias_corr` is ignored."""
    nseg = len(seg_end)
    if not (0 < nseg <= ADAM_MAX_SEG and len(seg_lr) == nseg and len(seg_wd) == nseg):
        raise L.SdeHipError(f"adam_step: {nseg} segments (at most {ADAM_MAX_SEG})")
    d = AdamDesc()
    for i in range(nseg):
        d.seg_end[i], d.seg_lr[i], d.seg_wd[i] = int(seg_end[i]), float(seg_lr[i]), float(seg_wd[i])
    d.nseg, d.decoupled_wd = nseg, int(bool(decoupled_wd))
    d.beta1, d.beta2, d.eps, d.bias_corr1, d.bias_corr2, d.grad_scale = beta1, beta2, eps, float(bias_corr[0]), float(bias_corr[1]), grad_scale
    d.scale_state = scale_state.data_ptr() if scale_state is not None else None
    d.beta1_d, d.beta2_d = float(beta1), float(beta2)
    if scale_state is not None and scale_state.numel() < 4:
        raise L.SdeHipError("adam_step: scale_state must hold 4 floats {loss_scale, found_inf, growth_tracker, applied_steps}")
    L.check(L.lib().sde_adam_step(L.ptr(p), L.ptr(g), L.ptr(m), L.ptr(v), p.numel(), ctypes.byref(d), L.stream()), "sde_adam_step")


def grad_check(g, scale_state):
    """scale_state[1] = 1 when any element of the flat fp32 gradient is inf / nan (device side, no host sync)."""
    L.check(L.lib().sde_grad_check(L.ptr(g), g.numel(), L.ptr(scale_state), L.stream()), "sde_grad_check")


def loss_scale_update(scale_state, growth_factor=2.0, backoff_factor=0.5, growth_interval=2000):
    """torch.cuda.amp.GradScaler.update() on the device scalar triple."""
    L.check(L.lib().sde_loss_scale_update(L.ptr(scale_state), growth_factor, backoff_factor, int(growth_interval), L.stream()), "sde_loss_scale_update")
