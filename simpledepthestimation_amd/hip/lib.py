"""Loader for the C-ABI library declared in include/sde_hip.h.

The product path has NO fallback: if libsde_hip.so is missing or a call fails, this raises.
"""
import ctypes
import os
from ctypes import POINTER, Structure, c_char_p, c_float, c_int, c_int32, c_size_t, c_void_p

import torch

_HERE = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
LIB_PATH = os.environ.get("SDE_HIP_LIB", os.path.join(_HERE, "libsde_hip.so"))   # override: A/B builds of the same ABI
MAX_CTX = 4
F32, BF16 = 0, 1


class SdeHipError(RuntimeError):
    pass


class PhotoDesc(Structure):
    _fields_ = [("A", c_void_p), ("ctx", c_void_p * MAX_CTX), ("pose", c_void_p * MAX_CTX), ("depth", c_void_p), ("K", c_void_p),
                ("B", c_int32), ("h", c_int32), ("w", c_int32), ("nctx", c_int32), ("automask", c_int32), ("reduce_mean", c_int32),
                ("sx", c_float), ("sy", c_float), ("ssim_w", c_float), ("C1", c_float), ("C2", c_float), ("clip_thr", c_void_p)]


_P, _I, _F = c_void_p, c_int, c_float
_PROTOS = {
    "sde_version": ([], c_int),
    "sde_resize": ([_P, _P, _I, _I, _I, _I, _I, _I, _P], c_int),
    "sde_pose_vec2mat": ([_P, _P, _I, _P], c_int),
    "sde_pose_vec2mat_bwd": ([_P, _P, _P, _I, _P], c_int),
    "sde_view_synthesis": ([_P, _P, _P, _P, _F, _F, _I, _I, _I, _I, _P, _P, _P, _P, _P, _P, _P], c_int),
    "sde_photo_num_blocks": ([_I, _I, _I, _I], c_int),
    "sde_photo_fwd": ([POINTER(PhotoDesc), POINTER(c_void_p), _P, _P, _P, _P, _F, _I, _P], c_int),
    "sde_photo_bwd": ([POINTER(PhotoDesc), POINTER(c_void_p), _P, _P, _F, _P, _I, _P, POINTER(c_void_p), _I, _P], c_int),
    "sde_smooth_num_blocks": ([_I, _I, _I], c_int),
    "sde_smooth_fwd": ([_P, _P, _I, _I, _I, _P, _P, _P, _P, _P, _F, _I, _P], c_int),
    "sde_smooth_bwd": ([_P, _P, _P, _P, _P, _F, _I, _I, _I, _P, _I, _P], c_int),
    "sde_silog_num_blocks": ([_I, _I, _I], c_int),
    "sde_silog_fwd": ([_P, _P, _I, _I, _I, _I, _I, _F, _P, _P, _P], c_int),
    "sde_silog_bwd": ([_P, _P, _P, _P, _F, _F, _I, _I, _I, _I, _I, _P, _I, _P], c_int),
}

_lib = None
PROFILE = None   # bench.py sets this to a list to collect (kind, flops, variant, start_event, end_event) per GEMM launch


def timed(kind, work, variant, call, meta=None):
    """Run `call` bracketed by events on the current stream when PROFILE is a list (bench.py's roofline pass).
    `work` = the launch's algorithmic FLOPs (GEMM kinds) or bytes (photo_* kinds)."""
    if PROFILE is None:
        return call()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    r = call()
    e1.record()
    PROFILE.append((kind, float(work), variant, e0, e1, meta))
    return r


def register_protos(protos):
    """Other binding modules (conv, norm, ...) add their prototypes here before first use."""
    _PROTOS.update(protos)
    if _lib is not None:
        _bind(_lib, protos)


def _bind(L, protos):
    for name, (args, res) in protos.items():
        fn = getattr(L, name)   # AttributeError => the .so does not export what the header declares: fail loudly
        fn.argtypes = args
        fn.restype = res


def available():
    return os.path.exists(LIB_PATH)


def lib():
    global _lib
    if _lib is None:
        if not os.path.exists(LIB_PATH):
            raise SdeHipError(f"{LIB_PATH} not found: build it with `python -c 'import __graft_entry__ as g; g.build()'` "
                              "(make -C simpledepthestimation_amd/csrc). There is no CPU fallback.")
        L = ctypes.CDLL(LIB_PATH)
        L.sde_last_error.restype = c_char_p
        _bind(L, _PROTOS)
        _lib = L
    return _lib


def check(rc, name):
    if rc != 0:
        raise SdeHipError(f"{name} failed ({rc}): {lib().sde_last_error().decode()}")


def ptr(t):
    """Device pointer of a contiguous CUDA tensor (None -> NULL)."""
    if t is None:
        return c_void_p(0)
    if not t.is_cuda:
        raise SdeHipError("simpledepthestimation_amd ops need CUDA (HIP) tensors; there is no CPU fallback")
    if not t.is_contiguous():
        raise SdeHipError("non-contiguous tensor passed to a HIP op")
    return c_void_p(t.data_ptr())


def stream():
    return c_void_p(torch.cuda.current_stream().cuda_stream)


_side = {}
SIDE_STREAM = os.environ.get("SDE_WGRAD_SIDE_STREAM", "1") != "0"
# SDE_LATE_JOIN=1 joins the side stream once per backward phase (at the reducer's flush) instead of after every layer.  Measured WORSE
# (10.39 vs 9.69 ms/step, Supervised R50): operands kept alive for the lagging GEMMs stop the allocator from recycling hot blocks, and the
# step's working set falls out of the 256 MB Infinity Cache.  Off by default; kept as an experiment knob.
LATE_JOIN = os.environ.get("SDE_LATE_JOIN", "0") == "1"
# Join a layer's weight-gradient GEMM (side stream) this many convolutions later instead of at the end of its own backward: the main stream
# runs on into the next layers' BatchNorm / data-gradient work instead of idling at the per-layer join (measured, ms/step: lag 0 9.54,
# 1 9.20, 2 9.01, 3 9.10, 5 9.21); operands of the lagging layers are held alive, so the working set grows by that many layers only.
JOIN_LAG = int(os.environ.get("SDE_JOIN_LAG", "2"))
# > 1: the weight-gradient GEMMs of that many consecutive layers are launched behind ONE fork of the side stream (fewer cross-stream edges in the
# captured graph: a trace of the replayed graph shows the GPU 99.5 % busy without the side stream and 90 % with a fork + join per layer)
# (measured, Supervised R50 ms/step: group 1 9.22, 2 8.85, 3 8.84, 4 8.91, 6 8.90; MonoDepth2-R18 6.52 -> 6.33)
WGRAD_GROUP = int(os.environ.get("SDE_WGRAD_GROUP", "3"))
GROUP_MAX_BYTES = int(float(os.environ.get("SDE_WGRAD_GROUP_MAX_MB", "128")) * (1 << 20))     # layers with more operand bytes than this fork alone
PACK_SPLIT = os.environ.get("SDE_PACK_SPLIT", "0") == "1"          # pack the data-gradient operands on the side stream during the forward pass
DEFER_MAX_BYTES = int(float(os.environ.get("SDE_DEFER_MAX_MB", "2")) * (1 << 20))     # weight-gradient slab stacks up to this size join the batched reduction


N_SIDE = max(1, int(os.environ.get("SDE_SIDE_STREAMS", "1")))      # weight-gradient GEMMs of consecutive layers alternate over this many streams
_side_rr = [0]


def side_stream(rotate=True):
    """Per-device helper stream(s): weight-gradient GEMMs run there concurrently with the data-gradient GEMM of the same layer (fork/join
    with stream / event waits, so they are captured into the step's hipGraph as parallel branches).  With SDE_SIDE_STREAMS > 1 consecutive
    layers alternate over the streams, so two weight-gradient GEMMs can be in flight at once."""
    d = torch.cuda.current_device()
    if d not in _side:
        _side[d] = [torch.cuda.Stream(device=d) for _ in range(N_SIDE)]
    if rotate:
        _side_rr[0] = (_side_rr[0] + 1) % N_SIDE
    return _side[d][_side_rr[0]]


def all_side_streams():
    d = torch.cuda.current_device()
    return list(_side.get(d, []))


def ptr_array(tensors):
    arr = (c_void_p * MAX_CTX)()
    for i, t in enumerate(tensors):
        arr[i] = t.data_ptr() if t is not None else 0
    return arr
