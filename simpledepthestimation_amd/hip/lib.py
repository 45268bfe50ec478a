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
F32, BF16, F16 = 0, 1, 2


class SdeHipError(RuntimeError):
    pass


class PhotoDesc(Structure):
    _fields_ = [("A", c_void_p), ("ctx", c_void_p * MAX_CTX), ("pose", c_void_p * MAX_CTX), ("depth", c_void_p), ("K", c_void_p),
                ("B", c_int32), ("h", c_int32), ("w", c_int32), ("nctx", c_int32), ("automask", c_int32), ("reduce_mean", c_int32),
                ("sx", c_float), ("sy", c_float), ("ssim_w", c_float), ("C1", c_float), ("C2", c_float), ("clip_thr", c_void_p)]


_P, _I, _F = c_void_p, c_int, c_float
_PROTOS = {
    "sde_version": ([], c_int),
    "sde_mark_time": ([_P, _P], c_int),
    "sde_wall_clock_khz": ([], c_int),
    "sde_store_u64": ([_P, ctypes.c_uint64, _P], c_int),
    "sde_resize": ([_P, _P, _I, _I, _I, _I, _I, _I, _P], c_int),
    "sde_pose_vec2mat": ([_P, _P, _I, _P], c_int),
    "sde_pose_vec2mat_bwd": ([_P, _P, _P, _I, _P], c_int),
    "sde_view_synthesis": ([_P, _P, _P, _P, _F, _F, _I, _I, _I, _I, _P, _P, _P, _P, _P, _P, _P], c_int),
    "sde_photo_num_blocks": ([_I, _I, _I, _I], c_int),
    "sde_photo_fwd": ([POINTER(PhotoDesc), POINTER(c_void_p), _P, _P, _P, _P, _F, _I, _P], c_int),
    "sde_photo_bwd": ([POINTER(PhotoDesc), POINTER(c_void_p), _P, _P, _F, _P, _I, _P, POINTER(c_void_p), _I, _P], c_int),
    "sde_photo_multi_fwd": ([_P, _I, _P, _P, _P, _P, _P], c_int),
    "sde_photo_multi_bwd": ([_P, _I, _P, _P, _P, _P, _P, _P, _P], c_int),
    "sde_photo_multi_pose_finalize": ([_P, _I, _P, _P, _P], c_int),
    "sde_mono_loss_fwd": ([_P, _I, _P, _P, _P, _P, _P, _P, _P, _P, _P, _P, _P, _P, _P], c_int),
    "sde_mono_loss_bwd": ([_P, _I, _P, _P, _P, _P, _P, _P, _P, _P, _P, _P, _P, _P, _P], c_int),
    "sde_smooth_multi_bwd": ([_P, _I, _P, _P, _P, _P, _P, _P, _I, _P], c_int),
    "sde_ssim_fwd": ([_P, _P, _I, _I, _I, _I, _F, _F, _P, _P], c_int),
    "sde_ssim_bwd": ([_P, _P, _P, _I, _I, _I, _I, _F, _F, _P, _P, _P, _P], c_int),
    "sde_smooth_num_blocks": ([_I, _I, _I], c_int),
    "sde_smooth_fwd": ([_P, _P, _I, _I, _I, _P, _P, _P, _P, _P, _F, _I, _P], c_int),
    "sde_smooth_bwd": ([_P, _P, _P, _P, _P, _F, _I, _I, _I, _P, _I, _P], c_int),
    "sde_silog_num_blocks": ([_I, _I, _I], c_int),
    "sde_silog_fwd": ([_P, _P, _I, _I, _I, _I, _I, _F, _P, _P, _P], c_int),
    "sde_silog_bwd": ([_P, _P, _P, _P, _F, _F, _I, _I, _I, _I, _I, _P, _I, _P], c_int),
    "sde_silog_multi_num_blocks": ([_I, _P, _P, _I], c_int),
    "sde_silog_multi_fwd": ([_P, _P, _I, _P, _P, _P, _I, _I, _I, _F, _P, _P, _P, _P], c_int),
    "sde_silog_multi_bwd": ([_P, _P, _P, _P, _F, _F, _I, _P, _P, _P, _I, _I, _I, _P, _P], c_int),
}

_lib = None
PROFILE = None   # bench.py sets this to a list to collect (kind, flops, variant, start_event, end_event, meta, repeat) per GEMM launch
PROFILE_REPEAT = 1


def timed(kind, work, variant, call, meta=None):
    """Run `call` bracketed by events on the current stream when PROFILE is a list (bench.py's roofline pass).
    `work` = the launch's algorithmic FLOPs (GEMM kinds) or bytes (photo_* kinds)."""
    if PROFILE is None:
        return call()
    # An event pair around ONE launch also times the launch path itself (8-12 us on this stack: the per-kernel durations of a rocprofv3 trace of
    # the same step were that much shorter), so the idempotent GEMM launches are issued PROFILE_REPEAT times back to back between the two events
    # and the record carries the count: duration = elapsed / repeat (what remains on top of rocprofv3's figure is one kernel boundary, ~1.5 us).
    rep = PROFILE_REPEAT if (kind.startswith("igemm") or kind == "wgrad") else 1
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(rep):
        r = call()
    e1.record()
    PROFILE.append((kind, float(work), variant, e0, e1, meta, rep))
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
# Weight-gradient GEMMs run on ONE side stream, forked in groups and joined late (hip/nn.py: _Conv2d.backward, WGradReducer).  The values below
# were each decided by an A/B run inside one gpurun call in round 1 (DESIGN.md 6.1 keeps the measurements); the environment switches that
# selected the losing variants (late join per phase, several side streams, packing on the side stream) were removed with those variants.
SIDE_STREAM = True      # HipTrainer(side_stream=False) / hip.lib.SIDE_STREAM = False: single-stream backward (profiling, debugging)
JOIN_LAG = 1            # the main stream joins all but the newest JOIN_LAG - 1 groups before it forks the next one
WGRAD_GROUP = 6         # weight-gradient GEMMs of that many consecutive layers behind one fork
# The two constants are set per network family when a HipTrainer is built (apply_schedule), from A/B runs inside one gpurun call each:
#   round 1 (register-staged weight gradients): lag 0 / 1 / 2 / 3 -> 9.54 / 9.20 / 9.01 / 9.10, group 1 / 2 / 3 / 4 -> 9.22 / 8.85 / 8.84 / 8.91 ms/step;
#   round 2 (LDS-DMA / halo weight gradients, faster slab sums: the side queue got a third shorter), Supervised-R50: (lag, group) = (2, 3) 7.21,
#   (1, 3) 7.09, (1, 4) 7.10, (1, 5) 7.06, (1, 6) 6.98-7.00, (1, 7) 7.10, (1, 8) 7.00, (1, 10) 7.07, (0, 6) 6.98, (0, 8) 7.00, (2, 8) 7.13;
#   MonoDepth2-R18 5.48 -> 5.33, R50 8.98 -> 8.74 with (1, 6); PackNet-1A the other way: (2, 3) 53.8, (1, 3) 55.9, (1, 6) 56.0 ms/step.
#   FIRST_GROUP (layers in the first group of a phase; until it closes nothing runs next to the chain; it also shifts every later group boundary):
#   bottleneck ResNets 0 (= 6) 6.77 against 6.89-6.94 ms/step for 1 / 2 / 3 / 4 / 5 / 7 / 8 / 9 (Supervised-R50), MonoDepth2-R50 8.06 vs 8.09 at 3;
#   basic-block ResNets 3: Supervised-R18 3.50 -> 3.42, MonoDepth2-R18 4.69 -> 4.60 (2: 4.62, 4: 4.74, 5: 4.66).
SCHEDULES = {"resnet": (1, 6, 0), "resnet_basic": (1, 6, 3), "packnet": (2, 3, 0)}
# ... and the workgroup target of the weight-gradient GEMMs' pixel splits (SDE_OPT_WGRAD_BLOCKS).  Round 3, one call: ResNet-50 256 / 384 / 512 / 768 -> 6.37 / 6.45 /
# 6.50 / 6.64 ms/step; PackNet-1A (weight gradients over 1.47 M pixels with 64 ... 512 channels) 256 / 512 / 768 / 1024 / 1536 / 2048 / 4096 -> 50.7-51.4 / 48.8 /
# 48.4 / 47.9-48.0 / 48.1 / 48.6 / 50.5 (profiles/r03ah_packnet_sweep.txt)
WGRAD_BLOCKS = {"resnet": 256, "resnet_basic": 256, "packnet": 1024}
WGRAD_BLOCKS_LOCKED = False      # bench.py --opt 6=... pins the value for an A/B run
SCHEDULE_LOCKED = False  # bench.py --const JOIN_LAG=... / WGRAD_GROUP=... pins the values for an A/B run


def apply_schedule(family):
    global JOIN_LAG, WGRAD_GROUP, FIRST_GROUP
    if not SCHEDULE_LOCKED:
        JOIN_LAG, WGRAD_GROUP, FIRST_GROUP = SCHEDULES[family]
FIRST_GROUP_B = None            # first-group digits of the SECOND phase of a two-phase (data-parallel) backward; None = FIRST_GROUP.  The group boundaries decide
                                # how much weight-gradient work is left behind the phase's last data-gradient layer (its exposed tail)
FIRST_GROUP = 0                 # sizes of the first groups of a backward phase as decimal digits (3: the first group has 3 layers; 33: the first two), 0: none;
                                # set per network family (SCHEDULES)
GROUP_MAX_BYTES = 128 << 20     # layers with more operand bytes fork alone (PackNet's 190 MB maps: 57.7 vs 60.2 ms/step)
GROUP_BUDGET_BYTES = 384 << 20  # ... and a group also closes once its layers' operands add up to this much
DEFER_MAX_BYTES = 2 << 20       # slab stacks up to this size join the phase's batched reduction; bigger ones are summed at once, cache-resident
FORK_MIN_BYTES = 0              # layers with fewer operand bytes keep their weight-gradient GEMM on the main stream (0: every layer forks; measured
                                # 10 / 25 / 50 / 100 MB -> 7.52 / 7.93 / 8.77 / 7.78 ms/step against 7.26-7.28 at 0: bench.py --const FORK_MIN_BYTES=...)


def side_stream(rotate=True):
    """Per-device helper stream: weight-gradient GEMMs run there concurrently with the data-gradient GEMM of the same layers (fork / join with
    stream / event waits, so they are captured into the step's hipGraph as a parallel branch)."""
    d = torch.cuda.current_device()
    if d not in _side:
        _side[d] = [torch.cuda.Stream(device=d)]
    return _side[d][0]


_aux = {}
# MonoDepth2: PoseNet (seven small layers, every kernel of it launch-bound at batch 12) runs on an auxiliary stream underneath the depth network's forward
# pass, and -- autograd replays a node's backward on the stream its forward ran on -- its backward underneath the depth network's backward.  HipTrainer
# switches this on for single-process training (the two-phase backward of the data-parallel path keeps one stream).
POSE_STREAM = False
AUX_USED = False        # set by the forward pass that forked the auxiliary stream; the trainer joins it after backward


def aux_stream():
    d = torch.cuda.current_device()
    if d not in _aux:
        _aux[d] = torch.cuda.Stream(device=d)
    return _aux[d]


def is_aux_stream(st):
    return any(st == a for a in _aux.values())


def join_aux():
    """Make the current stream wait for the auxiliary stream's work of this step (no-op when nothing was forked)."""
    global AUX_USED
    if AUX_USED:
        torch.cuda.current_stream().wait_stream(aux_stream())
        AUX_USED = False


def all_side_streams():
    d = torch.cuda.current_device()
    return list(_side.get(d, [])) + ([_aux[d]] if d in _aux else [])


# Timeline markers (bench.py --marks): mark(name) enqueues one sde_mark_time launch on the current stream; under graph capture it becomes a node of the
# step's graph, so every replay refreshes the slot.  MARKS is None unless a diagnostic run switched them on: mark() is then a no-op.
MARKS = None


def marks_enable(device):
    global MARKS
    MARKS = {"names": [], "slots": torch.zeros(128, dtype=torch.int64, device=device)}


def mark(name):
    if MARKS is None:
        return
    if name not in MARKS["names"]:
        MARKS["names"].append(name)
    i = MARKS["names"].index(name)
    check(lib().sde_mark_time(c_void_p(MARKS["slots"].data_ptr() + 8 * i), stream()), "sde_mark_time")


def marks_read(origin=None):
    """{name: microseconds since the `origin` marker (default: the earliest one)} of the last step's markers (synchronises)."""
    if MARKS is None:
        return {}
    torch.cuda.synchronize()
    khz = lib().sde_wall_clock_khz()
    t = MARKS["slots"][:len(MARKS["names"])].tolist()
    t0 = t[MARKS["names"].index(origin)] if origin in MARKS["names"] else min(t)
    return {n: (v - t0) * 1e3 / khz for n, v in zip(MARKS["names"], t)}


def ptr_array(tensors):
    arr = (c_void_p * MAX_CTX)()
    for i, t in enumerate(tensors):
        arr[i] = t.data_ptr() if t is not None else 0
    return arr
