"""Training step engine for the HIP path: flat parameter/gradient buffers, fused Adam/AdamW, whole-step hipGraph capture and
data-parallel gradient all-reduce over RCCL.

What it replaces in the reference: the body of the hot loop of projects/Supervised/train.py:L99-128 and
projects/MonoDepth2/train.py:L80-107 (``model(data)``, ``losses.backward()``, ``optimizer.step()``, LR schedule) and the
DistributedDataParallel wrapper of detectron2/utils/setup.py:L38-45.

MI355X-first choices
  * one process per GPU; every trainable tensor is a view into ONE flat fp32 buffer (parameters) with a twin flat gradient
    buffer, so the optimizer is a single bandwidth-bound kernel (sde_adam_step) and the gradient exchange is a handful of large
    RCCL all-reduces over xGMI instead of one per tensor;
  * zero-grad + forward + backward are captured once into a hipGraph (static shapes: 192x640, fixed batch) and replayed, which
    removes the per-kernel host launch cost (several hundred launches per step); collectives and the optimizer kernel run
    between replays on the same stream;
  * gradients are summed across ranks (all-reduce SUM) and the 1/world factor is applied inside the Adam kernel;
  * BatchNorm uses per-rank statistics and buffers are never re-broadcast (the reference passes broadcast_buffers=False).
"""
import math
import os

import numpy as np
import torch
import torch.distributed as dist

from ..hip import lib as L
from ..hip import nn as HN


def _dist_on():
    return dist.is_available() and dist.is_initialized() and dist.get_world_size() > 1


class ParamGroup:
    def __init__(self, name, named_params, lr, weight_decay=0.0):
        self.name, self.named_params, self.lr, self.weight_decay = name, list(named_params), float(lr), float(weight_decay)


def flat_view(buf, off, p):
    """The slice of a flat buffer that belongs to parameter p, shaped like p.  4-D (convolution) weights live in the buffers channels-last --
    memory order [Cout][KH][KW][Cin], the order of the packed MFMA operands and of the weight-gradient slabs, so the per-step weight pack is a
    cast and the slab reduction a plain sum -- under the usual [Cout,Cin,KH,KW] shape; state dicts, torch.optim state and gradient comparisons
    see logical tensors and never notice."""
    n = p.numel()
    if p.dim() == 4:
        Cout, Cin, KH, KW = p.shape
        return buf[off:off + n].view(Cout, KH, KW, Cin).permute(0, 3, 1, 2)
    return buf[off:off + n].view(p.shape)


def _slot(n):
    return (n + 3) & ~3


class GradCut:
    """Splits the autograd graph at a set of activations (the encoder features feeding layer3 and the decoder skips).

    forward: cut(feats) returns detached leaves that the "late" part of the network consumes; phase A's loss.backward() then stops at
    the leaves (filling leaf.grad) after producing every late parameter gradient; phase B continues from the original tensors.
    """

    def __init__(self):
        self.orig, self.leaves = None, None

    def __call__(self, feats):
        self.orig = list(feats)
        self.leaves = [f.detach().requires_grad_(True) for f in feats]
        return self.leaves

    def backward_rest(self):
        torch.autograd.backward(self.orig, [l.grad for l in self.leaves])


class HipTrainer:
    """Owns the flat buffers and runs one training step: loss dict = trainer.step(batch)."""

    def __init__(self, model, groups, adamw=False, betas=(0.9, 0.999), eps=1e-8, bucket_mb=64, use_graph=False, skip_unused=(".fc.",),
                 adam_fn=None, overlap=None, late_from=("layer3",), cut_owner=None, amp=False, init_scale=65536.0, growth_interval=2000, pose_stream=True, cu_reserve=None):
        self.model = model
        kinds = {type(m).__name__ for m in model.modules()}
        family = "packnet" if "PackNet01" in kinds else "resnet" if ("Bottleneck" in kinds or "BasicBlock" not in kinds) else "resnet_basic"
        L.apply_schedule(family)
        if not L.WGRAD_BLOCKS_LOCKED and next(model.parameters()).device.type == "cuda":
            HN.set_option(HN.OPT_WGRAD_BLOCKS, L.WGRAD_BLOCKS[family])
        self._adam_fn = adam_fn or HN.adam_step      # tests on CPU (gloo) substitute a torch restatement of the same update
        self.adamw, self.betas, self.eps = bool(adamw), betas, float(eps)
        self.use_graph = bool(use_graph)
        self.world = dist.get_world_size() if _dist_on() else 1
        dev = next(model.parameters()).device
        self.device = dev
        # ---- flatten: parameters of each group become views of one buffer, in group order
        self.groups = []
        self._all_names = []           # per group: EVERY parameter name in registration order, incl. the skipped ones (torchvision's unused fc):
        total = 0                      # a torch.optim state dict of the reference numbers its parameters over that full list
        for g in groups:
            kept = [(n, p) for n, p in g.named_params if p.requires_grad and not any(s in n for s in skip_unused)]
            self.groups.append(ParamGroup(g.name, kept, g.lr, g.weight_decay))
            self._all_names.append([n for n, _ in g.named_params])
            total += sum(_slot(p.numel()) for _, p in kept)
        self.numel = total             # every parameter starts on a 16-byte boundary of the flat buffers (float4 kernels); the padding stays zero
        self.pflat = torch.zeros(total, device=dev, dtype=torch.float32)
        self._off = {}                 # id(parameter) -> offset of its slot
        self.gflat = torch.zeros(total, device=dev, dtype=torch.float32)
        self.m = torch.zeros(total, device=dev, dtype=torch.float32)
        self.v = torch.zeros(total, device=dev, dtype=torch.float32)
        off, seg_end = 0, []
        self._module_cuts = set()      # offsets where the top-level sub-module changes (encoder.layer3 | encoder.layer4 | decoder ...): bucket boundaries
        prev_mod = None
        for g in self.groups:
            for n_, p in g.named_params:
                parts = n_.split(".")
                i_layer = next((i for i, t in enumerate(parts) if t.startswith("layer") or t in ("decoder", "pose_net", "conv1", "bn1")), None)
                mod = (g.name, ".".join(parts[:i_layer] + ["stem" if parts[i_layer] in ("conv1", "bn1") else parts[i_layer]]) if i_layer is not None else parts[0])
                if mod != prev_mod:
                    self._module_cuts.add(off)
                    prev_mod = mod
                self._off[id(p)] = off
                flat_view(self.pflat, off, p).copy_(p.data)
                p.data = flat_view(self.pflat, off, p)
                p.grad = flat_view(self.gflat, off, p)
                off += _slot(p.numel())
            seg_end.append(off)
        self.seg_end = seg_end         # host tables: the optimizer kernel takes them by value (no upload to order against the launch)
        self.t = 0
        # fp16 training (SOLVER.AMP, detectron2/engine/train_loop.py:L294-341): device-side GradScaler state {scale, found_inf, growth tracker}
        self.amp = bool(amp)
        self.growth_interval = int(growth_interval)
        # + the count of APPLIED optimizer steps: GradScaler.step skips optimizer.step() on overflow, so Adam's `step` (bias corrections) must not
        # advance then -- the Adam kernel derives its bias corrections from this device-side count, self.t only counts calls
        self.scale_state = torch.tensor([float(init_scale), 0.0, 0.0, 0.0], device=dev, dtype=torch.float32) if self.amp else None
        # ---- gradient buckets for the all-reduce (contiguous slices of the flat gradient)
        per = max(1, int(bucket_mb * (1 << 20) / 4))
        self.buckets = [(s, min(total, s + per)) for s in range(0, total, per)]
        if self.world > 1:
            dist.broadcast(self.pflat, src=0)          # DDP's initial parameter broadcast (setup.py:L40)
        # ---- two-phase backward (overlap the all-reduce of the "late" parameters with the rest of backward)
        owner = cut_owner if cut_owner is not None else getattr(model, "depth_net", None)
        self.overlap = (self.world > 1) if overlap is None else bool(overlap)
        self._cut, self.late_start = None, total
        if self.overlap and owner is not None and hasattr(owner, "_grad_cut"):
            off = 0
            for g in self.groups:
                for n, p in g.named_params:
                    if self.late_start == total and any(t in n for t in late_from):
                        self.late_start = self._off[id(p)]
            if self.late_start < total:
                self._cut = GradCut()
                owner._grad_cut = self._cut
        if self._cut is None:
            self.overlap = False
        # MonoDepth2: PoseNet on the auxiliary stream underneath the depth network (hip/lib.py: POSE_STREAM).  Under the two-phase backward of the
        # data-parallel path as well: PoseNet's backward does not depend on the cut features, so all of it runs in phase A on the auxiliary
        # stream and _backward joins that stream before phase A's slab reduction -- its gradients are final when the late all-reduce starts
        self.pose_stream = bool(pose_stream) and adam_fn is None
        # data-parallel runs: compute units the persistent kernels leave to RCCL's channel kernels (SDE_OPT_CU_RESERVE).  Off unless asked for: its
        # effect can only be measured on a multi-GPU node (SDE_CU_RESERVE=<multiple of 8>, or HipTrainer(cu_reserve=...))
        if cu_reserve is None:
            cu_reserve = int(os.environ.get("SDE_CU_RESERVE", "0")) if self.world > 1 else 0
        self.cu_reserve = int(cu_reserve)
        if self.cu_reserve and dev.type == "cuda":
            HN.set_option(HN.OPT_CU_RESERVE, self.cu_reserve)
        self._graph_b = None
        self._graph = None
        self._graphs = {}
        self._static_batch = None
        self._static_out = None
        # optional callable(batch, static_batch_or_None) -> batch run at the head of every step on the step's stream: data/device_aug.py's DeviceImageAug
        # turns uploaded uint8 frames into the fp32 image entries there, writing straight into the captured graph's static inputs
        self.input_transform = None
        self._packer = None            # built lazily after the first eager step (needs the operand shapes seen in forward)
        self._one = None
        self.batch_pack = adam_fn is None
        # one weight-gradient slab reduction launch per backward phase
        self._wreduce = HN.WGradReducer() if adam_fn is None else None

    # ------------------------------------------------------------------------------------------------------------
    def set_lr(self, lrs):
        """lrs: one value per group.  Host-side only: the next optimizer launch carries the values in its kernel arguments."""
        if len(lrs) != len(self.groups):
            raise ValueError(f"set_lr: {len(lrs)} values for {len(self.groups)} parameter groups")
        for g, lr in zip(self.groups, lrs):
            g.lr = float(lr)

    # ---- optimizer state in torch.optim.Adam/AdamW's state_dict layout, so that it travels through the reference's checkpoints
    # (DetectionCheckpointer(model, dir, optimizer=...) of projects/*/train.py): parameters are numbered group by group in
    # registration order; every parameter carries {'step', 'exp_avg', 'exp_avg_sq'}.  'param_names' is an addition that lets a load
    # match by name when the numbering differs (the reference's encoder group also holds the unused fc.weight / fc.bias).
    def applied_steps(self):
        """Optimizer steps actually applied: under AMP the device-side count (skipped overflow steps excluded; one host read), else self.t."""
        return int(self.scale_state[3].item()) if self.amp else self.t

    def state_dict(self):
        state, groups, idx = {}, [], 0
        t_applied = self.applied_steps()
        for g in self.groups:
            ids = []
            for _, p in g.named_params:
                off = self._off[id(p)]
                if t_applied > 0:
                    state[idx] = {"step": torch.tensor(float(t_applied)), "exp_avg": flat_view(self.m, off, p).detach().cpu().contiguous().clone(),
                                  "exp_avg_sq": flat_view(self.v, off, p).detach().cpu().contiguous().clone()}
                ids.append(idx)
                idx += 1
            groups.append({"lr": g.lr, "betas": tuple(self.betas), "eps": self.eps, "weight_decay": g.weight_decay, "amsgrad": False,
                           "params": ids, "param_names": [n for n, _ in g.named_params], "name": g.name})
        out = {"state": state, "param_groups": groups}
        if self.amp:                                   # GradScaler.state_dict(): the one host read of the scale, at checkpoint time
            sc = self.scale_state.detach().cpu().tolist()
            out["loss_scale"], out["growth_tracker"] = sc[0], sc[2]
        return out

    def load_state_dict(self, sd):
        """Accepts what state_dict() writes and what torch.optim.Adam(W).state_dict() of the reference's optimizer writes (same groups,
        same order).  Within a group parameters are matched by 'param_names' when present, otherwise by their position in the group's FULL
        registration order -- the reference's groups also hold torchvision's unused fc.weight / fc.bias (in the middle of MonoDepth2's
        'Depth' group), which this trainer skips but which still occupy positions in the file.  Everything is validated before the first
        write: a mismatch raises ValueError and leaves the moments untouched."""
        their_groups = sd["param_groups"]
        if len(their_groups) != len(self.groups):
            raise ValueError(f"optimizer checkpoint has {len(their_groups)} parameter groups, this trainer {len(self.groups)}")
        state = {int(k): v for k, v in sd["state"].items()}
        plan, steps = [], []
        for g, tg, all_names in zip(self.groups, their_groups, self._all_names):
            names = tg.get("param_names")
            ids = list(tg["params"])
            if names:
                by_name = dict(zip(names, ids))
            elif len(ids) == len(all_names):
                by_name = dict(zip(all_names, ids))                       # the reference's numbering: every registered parameter, fc included
            elif len(ids) == len(g.named_params):
                by_name = dict(zip([n for n, _ in g.named_params], ids))    # a file written over the kept parameters only
            else:
                raise ValueError(f"optimizer group '{g.name}': checkpoint lists {len(ids)} parameters, the model registers {len(all_names)} "
                                 f"({len(g.named_params)} of them trained)")
            for n, p in g.named_params:
                off = self._off[id(p)]
                ent = state.get(by_name.get(n))
                if ent is not None:
                    for key in ("exp_avg", "exp_avg_sq"):
                        if tuple(ent[key].shape) != tuple(p.shape):
                            raise ValueError(f"optimizer state {by_name.get(n)} ({key}) has shape {tuple(ent[key].shape)}, parameter '{n}' {tuple(p.shape)}")
                    plan.append((off, p, ent))
                    steps.append(int(float(ent["step"])))
        if steps and min(steps) != max(steps):
            raise ValueError("per-parameter step counts differ; the fused Adam kernel keeps one step count for all parameters")
        # ---- validated: now write
        self.m.zero_()
        self.v.zero_()
        for off, p, ent in plan:
            flat_view(self.m, off, p).copy_(ent["exp_avg"].to(self.m))
            flat_view(self.v, off, p).copy_(ent["exp_avg_sq"].to(self.v))
        for g, tg in zip(self.groups, their_groups):
            g.lr, g.weight_decay = float(tg.get("lr", g.lr)), float(tg.get("weight_decay", g.weight_decay))
        self.t = steps[0] if steps else 0
        if self.amp and "loss_scale" in sd:
            self.scale_state.copy_(torch.tensor([float(sd["loss_scale"]), 0.0, float(sd.get("growth_tracker", 0.0)), float(self.t)]))
        elif self.amp:
            self.scale_state[3] = float(self.t)

    def _fwd_bwd(self, batch):
        L.mark("step_start")
        self.gflat.zero_()
        if self._packer is not None:
            self._packer.run()                      # every conv operand of the step in one launch
        L.POSE_STREAM = self.pose_stream and self.device.type == "cuda"
        try:
            out = self.model(batch)
        finally:
            L.POSE_STREAM = False
        loss_dict = {k: v for k, v in out.items() if "loss" in k}
        terms = list(loss_dict.values())
        losses = terms[0]                               # (python's sum() starts from int 0: an add-scalar kernel on the serial seam of the step)
        for t in terms[1:]:
            losses = losses + t
        if self.amp:
            losses = losses * self.scale_state[0]       # GradScaler.scale(loss): a device scalar, so the multiply is part of the captured graph
        if self._packer is not None:
            self._packer.join_dgrad()
        if self._one is None or self._one.dtype != losses.dtype or self._one.device != losses.device or self._one.shape != losses.shape:
            self._one = torch.ones_like(losses)         # d loss / d loss, kept: backward() would launch a fill for it every step
        L.mark("loss_fwd_end")
        self._backward(lambda: losses.backward(gradient=self._one), first_of_two=self._cut is not None)
        L.mark("bwd_joined")
        if self._packer is None and self.batch_pack:
            try:
                self._packer = HN.WeightPacker(self.model)
            except Exception:
                self.batch_pack = False             # model without HIP convolutions
        return loss_dict

    def _backward(self, run, first_of_two=False):
        """One backward phase with the convolutions' weight-gradient reductions deferred to a single launch at its end."""
        HN.WGRAD_DEFER = self._wreduce
        # the fork / lagging-join bookkeeping of the weight-gradient side stream (hip/nn.py) is written against ONE main stream: every
        # convolution's backward must run on the stream this phase started on
        HN.MAIN_STREAM = torch.cuda.current_stream() if self.device.type == "cuda" else None
        tag = "" if self._cut is None else ("A_" if first_of_two else "B_")       # (marker names per phase of a two-phase backward)
        if self._wreduce is not None:
            self._wreduce.first_group = L.FIRST_GROUP_B if (self._cut is not None and not first_of_two) else None
        try:
            L.mark(tag + "bwd_start")
            run()
            if L.MARKS is not None:                  # diagnostic timeline (bench.py --marks): the last work of each stream of the phase
                L.mark(tag + "bwd_main_end")
                capturing = torch.cuda.is_current_stream_capturing()
                for name, st_ in [(tag + "bwd_side_end", L.side_stream())] + ([(tag + "bwd_aux_end", L.aux_stream())] if L.AUX_USED else []):
                    with torch.cuda.stream(st_):
                        if torch.cuda.is_current_stream_capturing() == capturing:      # (a helper stream this phase never forked is not part of the capture)
                            L.mark(name)
                            joined = True
                        else:
                            joined = False
                    if joined:
                        torch.cuda.current_stream().wait_stream(st_)                   # the marker is the stream's last node: join it
            L.join_aux()            # a network that ran on the auxiliary stream (PoseNet): its backward ran there as well
            if self._wreduce is not None:
                self._wreduce.flush()
        finally:
            HN.WGRAD_DEFER = None
            HN.MAIN_STREAM = None
            HN._RES_GRAD.clear(); HN._BN_PART.clear()      # hand-over entries between backward nodes: nothing outlives the phase
            L.join_aux()            # (backward raised before the join above)
            if self._wreduce is not None:
                self._wreduce.join_pending()
                if self._wreduce.forked:                                  # backward raised before the flush: still join the side stream
                    for st_ in HN.L.all_side_streams():
                        torch.cuda.current_stream().wait_stream(st_)
                    self._wreduce.forked = False
                self._wreduce.jobs, self._wreduce._seen = [], set()       # nothing left registered if backward raised
                self._wreduce.bias_jobs, self._wreduce._seen_bias = [], set()
                self._wreduce.queue, self._wreduce.groups_done = [], 0
            L.mark(tag + "bwd_end")

    def _backward_rest(self):
        self._backward(self._cut.backward_rest)

    def bucket_ranges(self, lo=0, hi=None, reverse=False):
        """The slices of gflat[lo:hi] that travel as one all-reduce each: cut at the parameter-group boundaries and at the top-level module
        boundaries inside a group (decoder | layer4 | layer3 | ...), then at bucket_mb.  reverse=True yields them last-created first -- the order in
        which backward finalises them and the order DDP's reducer sends its buckets (detectron2/utils/setup.py:L38-45 wraps the reference in DDP)."""
        hi = self.numel if hi is None else hi
        per = self.buckets[0][1] - self.buckets[0][0]
        cuts = sorted({c for c in self._module_cuts if lo < c < hi} | {lo, hi})
        out = []
        for a, b in zip(cuts, cuts[1:]):
            out += [(s, min(b, s + per)) for s in range(a, b, per)]
        return out[::-1] if reverse else out

    def _allreduce(self, lo=0, hi=None, wait=True, reverse=False):
        """SUM all-reduce of gflat[lo:hi] in buckets; returns the work handles (already waited on unless wait=False)."""
        if self.world == 1:
            return []
        handles = []
        for a, b in self.bucket_ranges(lo, hi, reverse):
            if a < b:
                handles.append(dist.all_reduce(self.gflat[a:b], op=dist.ReduceOp.SUM, async_op=True))
        if wait:
            for h in handles:
                h.wait()
        return handles

    def _optimizer(self):
        self.t += 1
        b1, b2 = self.betas
        bias_corr = (1.0 - b1 ** self.t, 1.0 - b2 ** self.t)       # by value: no per-step device scalar to keep ordered with the launch
        lrs, wds = [g.lr for g in self.groups], [g.weight_decay for g in self.groups]
        if self.amp:
            # GradScaler.step + update without a host sync: the check raises found_inf on the device, the optimizer kernel divides by the
            # scale and skips itself on overflow, the update kernel backs the scale off or counts towards growth
            HN.grad_check(self.gflat, self.scale_state)
            self._adam_fn(self.pflat, self.gflat, self.m, self.v, self.seg_end, lrs, wds, bias_corr, b1, b2, self.eps, 1.0 / self.world, self.adamw,
                          scale_state=self.scale_state)
            HN.loss_scale_update(self.scale_state, 2.0, 0.5, self.growth_interval)
        else:
            self._adam_fn(self.pflat, self.gflat, self.m, self.v, self.seg_end, lrs, wds, bias_corr, b1, b2, self.eps, 1.0 / self.world, self.adamw)

    # ------------------------------------------------------------------------------------------------------------
    # ---- hipGraph path: the batch dicts of the reference's collate (data/datasets/kitti_v2.py:L196-221) hold, besides tensors, numpy
    # arrays (ctx_img / ctx_img_orig lists), python scalars that steer the model (`flip`: ONE bool per batch) and per-sample
    # bookkeeping the kernels never read (`metadata`: list of dicts, file names).  Arrays become device tensors with static addresses;
    # steering scalars select the graph set (one capture per distinct value, lazily); everything else passes through by reference.
    @staticmethod
    def _is_array(v):
        return torch.is_tensor(v) or isinstance(v, np.ndarray)

    @classmethod
    def _kind(cls, v):
        if cls._is_array(v):
            return "array"
        if isinstance(v, (list, tuple)) and len(v) > 0 and all(cls._is_array(x) for x in v):
            return "arrays"
        if isinstance(v, (bool, int, float, str, type(None), np.bool_, np.integer, np.floating)):
            return "scalar"
        return "opaque"

    def _graph_key(self, batch):
        return tuple(sorted((k, v.item() if isinstance(v, np.generic) else v) for k, v in batch.items() if self._kind(v) == "scalar"))

    def _to_static(self, batch):
        def conv(v):
            return torch.as_tensor(v).to(self.device).clone()
        out = {}
        for k, v in batch.items():
            kind = self._kind(v)
            out[k] = conv(v) if kind == "array" else ([conv(x) for x in v] if kind == "arrays" else v)
        return out

    def _staged(self, v):
        """The static batch may still be read by the previous replay: never write pageable host data straight into it.  Pageable
        tensors first land in a fresh device tensor (blocking copy), the copy into the static buffer is then device-to-device in
        stream order; pinned and device tensors are already stream-ordered."""
        v = torch.as_tensor(v)
        if v.device.type == "cpu" and not v.is_pinned():
            return v.to(self.device)
        return v

    def _copy_into_static(self, batch):
        if set(batch.keys()) != set(self._static_batch.keys()):
            raise RuntimeError(f"batch keys changed since capture: {sorted(batch.keys())} vs {sorted(self._static_batch.keys())}")
        for k, v in batch.items():
            s, kind = self._static_batch[k], self._kind(v)
            if kind == "array":
                if torch.is_tensor(v) and v is s:
                    continue                       # written in place by input_transform
                s.copy_(self._staged(v), non_blocking=True)
            elif kind == "arrays":
                if len(s) != len(v):
                    raise RuntimeError(f"batch entry '{k}' changed length ({len(s)} -> {len(v)})")
                for a, b in zip(s, v):
                    if not (torch.is_tensor(b) and b is a):
                        a.copy_(self._staged(b), non_blocking=True)
            else:
                self._static_batch[k] = v          # steering scalars are part of the graph key; opaque entries are not read by captured work

    def capture(self, batch, warmup=3):
        """Warm up eagerly (sets kernel attributes, fills the allocator), then capture zero-grad + forward + backward for this batch's
        steering scalars (e.g. flip = False); other values get their own graph set on first use, over the same static input tensors."""
        if self._static_batch is None:
            self._static_batch = self._to_static(batch)
            self._graphs = {}
        else:
            self._copy_into_static(batch)
        key = self._graph_key(batch)
        # the warm-up steps and the capture pass must leave no trace in the model: BatchNorm running statistics (updated by every
        # training-mode forward) and the host-side batch counters are put back afterwards, so step k of a captured run sees exactly
        # the buffers step k of an eager run sees (and a resumed run continues from what the checkpoint held)
        self._bns = [m for m in self.model.modules() if hasattr(m, "_pending_batches")]
        kept_buffers = [b.detach().clone() for b in self.model.buffers()]
        kept_counts = [m._pending_batches for m in self._bns]
        s = torch.cuda.Stream()
        s.wait_stream(torch.cuda.current_stream())
        with torch.cuda.stream(s):
            for _ in range(warmup):
                self._fwd_bwd(dict(self._static_batch))
                if self._cut is not None:
                    self._backward_rest()
        torch.cuda.current_stream().wait_stream(s)
        torch.cuda.synchronize()
        graph, graph_b = torch.cuda.CUDAGraph(), None
        dot = os.environ.get("SDE_GRAPH_DOT")      # debugging aid: the captured DAG (kernel nodes + edges) as a Graphviz file
        if dot:
            graph.enable_debug_mode()
        # thread-local capture mode: other threads of the process keep making HIP calls while the step is captured -- RCCL's watchdog thread once the
        # process group exists (N > 1), a data loader's pin-memory thread -- and in the default "global" mode any of them invalidates the capture
        with torch.cuda.graph(graph, capture_error_mode="thread_local"):
            static_out = self._fwd_bwd(dict(self._static_batch))
        if dot:
            graph.debug_dump(dot)
        if self._cut is not None:                  # phase B: the rest of backward, same memory pool, replayed after phase A
            graph_b = torch.cuda.CUDAGraph()
            with torch.cuda.graph(graph_b, pool=graph.pool(), capture_error_mode="thread_local"):
                self._backward_rest()
        with torch.no_grad():
            for b, k in zip(self.model.buffers(), kept_buffers):
                b.copy_(k)
        for m, k in zip(self._bns, kept_counts):
            m._pending_batches = k
        self._graphs[key] = (graph, graph_b, static_out)
        self._graph, self._graph_b, self._static_out = graph, graph_b, static_out
        return self

    def step(self, batch):
        """One optimisation step.  Returns {loss name: 0-d device tensor} (no host sync)."""
        if self.input_transform is not None:
            L.mark("pre_transform")
            batch = self.input_transform(batch, None, self._static_batch if self.use_graph else None)
            L.mark("post_transform")
        if self.use_graph:
            hit = self._graphs.get(self._graph_key(batch)) if self._static_batch is not None else None
            if hit is None:
                self.capture(batch)                 # first batch, or a steering value (flip) not seen before; leaves the batch in the static buffers
            else:
                self._graph, self._graph_b, self._static_out = hit
                self._copy_into_static(batch)
            self._graph.replay()
            L.mark("post_replay")
            for m in self._bns:                     # the replay runs no Python forward: count the batch for num_batches_tracked here
                m._pending_batches += 1
            loss_dict = self._static_out
        else:
            loss_dict = self._fwd_bwd(batch)
        if self._cut is None:
            self._allreduce()
        else:
            # phase A is done: every gradient from late_start on is final -> reduce it while phase B (high-resolution layers) runs
            # in reverse creation order (decoder, then layer4, then layer3: the order backward finished them), one all-reduce per module-aligned bucket
            late = self._allreduce(self.late_start, self.numel, wait=False, reverse=True)
            if self._graph_b is not None:
                self._graph_b.replay()
            else:
                self._backward_rest()
            early = self._allreduce(0, self.late_start, wait=False, reverse=True)
            if os.environ.get("SDE_DIST_DEBUG"):          # debugging aid: host time spent waiting for each bucket's collective
                import time
                ts = []
                for h in late + early:
                    t = time.perf_counter(); h.wait(); ts.append(round((time.perf_counter() - t) * 1e3, 1))
                if self.rank == 0 if hasattr(self, "rank") else dist.get_rank() == 0:
                    print(f"[dist] waits ms late {ts[:len(late)]} early {ts[len(late):]} ranges late {self.bucket_ranges(self.late_start, self.numel, True)} "
                          f"early {self.bucket_ranges(0, self.late_start, True)}", flush=True)
            for h in late + early:
                h.wait()
        self._optimizer()
        L.mark("post_optimizer")
        return loss_dict


def _amp_kw(cfg, kw):
    if bool(cfg.SOLVER.get("AMP", False)):
        if str(cfg.MODEL.get("COMPUTE_DTYPE", "fp32")) != "fp16":
            raise ValueError("SOLVER.AMP needs MODEL.COMPUTE_DTYPE fp16 (bf16 and fp32 need no loss scaling)")
        kw.setdefault("amp", True)
    return kw


def supervised_trainer(model, cfg, **kw):
    """AdamW(enc wd 1e-2, dec wd 0, lr DEPTH_LR, eps 1e-6) -- projects/Supervised/train.py:L77-81."""
    m = model.module if hasattr(model, "module") else model
    groups = [ParamGroup("encoder", m.depth_net.encoder.named_parameters(prefix="depth_net.encoder"), cfg.SOLVER.DEPTH_LR, 1e-2),
              ParamGroup("decoder", m.depth_net.decoder.named_parameters(prefix="depth_net.decoder"), cfg.SOLVER.DEPTH_LR, 0.0)]
    return HipTrainer(m, groups, adamw=True, eps=1e-6, **_amp_kw(cfg, kw))


def monodepth2_trainer(model, cfg, **kw):
    """Adam(depth lr, pose lr, wd 0) -- projects/MonoDepth2/train.py:L50-57."""
    m = model.module if hasattr(model, "module") else model
    groups = [ParamGroup("Depth", m.depth_net.named_parameters(prefix="depth_net"), cfg.SOLVER.DEPTH_LR, 0.0),
              ParamGroup("Pose", m.pose_net.named_parameters(prefix="pose_net"), cfg.SOLVER.POSE_LR, 0.0)]
    return HipTrainer(m, groups, adamw=False, eps=1e-8, **_amp_kw(cfg, kw))


def poly_lr(cfg, global_step, max_iter):
    """projects/Supervised/train.py:L125-128."""
    return (cfg.SOLVER.DEPTH_LR - cfg.SOLVER.DEPTH_END_LR) * (1 - global_step / max_iter) ** 0.9 + cfg.SOLVER.DEPTH_END_LR


def multistep_lr(base_lr, epoch, milestones, gamma):
    """torch.optim.lr_scheduler.MultiStepLR as used by projects/MonoDepth2/train.py:L60-62,L109."""
    return base_lr * gamma ** sum(1 for m in milestones if epoch >= m)
