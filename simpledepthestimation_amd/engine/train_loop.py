"""Hook-driven trainers (reference: detectron2/engine/train_loop.py:L18-341 HookBase / TrainerBase / SimpleTrainer / AMPTrainer).

Same control flow and attribute names (iter, start_iter, max_iter, storage, register_hooks, train, run_step, before/after hooks), so hooks
written against the reference run unchanged.  What a step IS differs: SimpleTrainer.run_step hands the batch to a HipTrainer, whose step is
zero-grad + forward + backward (+ RCCL all-reduce) + fused Adam, replayed from the captured hipGraph; the `optimizer` attribute the
reference's hooks look for (LRScheduler, checkpointer) is that HipTrainer (it has param_groups-like groups, state_dict, set_lr).
AMPTrainer is the same loop over a HipTrainer built with amp=True: loss scaling, the overflow check, the skipped update and the scale
update all happen on the device (no GradScaler object, no host sync)."""
import logging
import time
import weakref

import numpy as np
import torch
import torch.distributed as dist

from ..utils.events import EventStorage, get_event_storage

__all__ = ["HookBase", "TrainerBase", "SimpleTrainer", "AMPTrainer"]


class HookBase:
    """before_train / after_train / before_step / after_step; `self.trainer` is a weak proxy set at registration (train_loop.py:L18-75)."""

    def before_train(self):
        pass

    def after_train(self):
        pass

    def before_step(self):
        pass

    def after_step(self):
        pass


class TrainerBase:
    def __init__(self):
        self._hooks = []
        self.iter = 0
        self.start_iter = 0
        self.max_iter = 0
        self.storage = None

    def register_hooks(self, hooks):
        hooks = [h for h in hooks if h is not None]
        for h in hooks:
            assert isinstance(h, HookBase)
            h.trainer = weakref.proxy(self)          # hooks and trainer must not own each other
        self._hooks.extend(hooks)

    def train(self, start_iter, max_iter):
        logger = logging.getLogger(__name__)
        logger.info("Starting training from iteration %d", start_iter)
        self.iter = self.start_iter = start_iter
        self.max_iter = max_iter
        with EventStorage(start_iter) as self.storage:
            try:
                self.before_train()
                for self.iter in range(start_iter, max_iter):
                    self.before_step()
                    self.run_step()
                    self.after_step()
                self.iter += 1          # iter == max_iter tells after_train that training finished (train_loop.py:L140-143)
            except Exception:
                logger.exception("Exception during training:")
                raise
            finally:
                self.after_train()

    def before_train(self):
        for h in self._hooks:
            h.before_train()

    def after_train(self):
        self.storage.iter = self.iter
        for h in self._hooks:
            h.after_train()

    def before_step(self):
        self.storage.iter = self.iter       # invariant: storage.iter == trainer.iter during a step
        for h in self._hooks:
            h.before_step()

    def after_step(self):
        for h in self._hooks:
            h.after_step()

    def run_step(self):
        raise NotImplementedError


class SimpleTrainer(TrainerBase):
    """model + data loader + optimizer, one step per iteration (train_loop.py:L181-291).  `optimizer` is a HipTrainer over `model`
    (engine.trainer.supervised_trainer / monodepth2_trainer build one); `model` may be the FakeDDP wrapper or the bare module."""

    def __init__(self, model, data_loader, optimizer):
        super().__init__()
        core = model.module if hasattr(model, "module") and isinstance(model.module, torch.nn.Module) else model
        core.train()
        if not (hasattr(optimizer, "step") and hasattr(optimizer, "set_lr") and getattr(optimizer, "model", None) is core):
            raise TypeError("SimpleTrainer: `optimizer` must be the HipTrainer built over `model` (engine.trainer.*_trainer)")
        self.model = model
        self.data_loader = data_loader
        self._data_loader_iter = iter(data_loader)
        self.optimizer = optimizer

    def run_step(self):
        core = self.optimizer.model
        assert core.training, "[SimpleTrainer] model was changed to eval mode!"
        start = time.perf_counter()
        try:
            data = next(self._data_loader_iter)
        except StopIteration:
            # a finite loader (SAMPLER_TRAIN "DDPSampler": one epoch per iterator) under the iteration-driven loop: start the next epoch.  The
            # reference pairs this trainer with the infinite TrainingSampler (data/build.py:L108-109), which build_detection_train_loader also builds
            self._epoch = getattr(self, "_epoch", 0) + 1
            smp = getattr(getattr(self.data_loader, "batch_sampler", None), "sampler", None)
            if hasattr(smp, "set_epoch"):
                smp.set_epoch(self._epoch)
            self._data_loader_iter = iter(self.data_loader)
            data = next(self._data_loader_iter)
        data_time = time.perf_counter() - start
        loss_dict = self.optimizer.step(data)          # zero-grad + forward + backward (+ all-reduce) + Adam: one graph replay
        self._write_metrics(loss_dict, data_time)

    def _write_metrics(self, loss_dict, data_time, prefix=""):
        """The reference gathers python floats from every worker each iteration (train_loop.py:L250-291).  Here the 0-d device tensors are
        averaged over the ranks by one small all-reduce (when N > 1) and parked in the storage as tensors; the finiteness check runs
        where they are first read (check_finite(), called by the writers' hook)."""
        names = sorted(loss_dict)
        vec = torch.stack([loss_dict[k].detach().float() for k in names])
        dt = data_time
        if dist.is_available() and dist.is_initialized() and dist.get_world_size() > 1:
            vec = vec.clone()
            dist.all_reduce(vec)
            vec = vec / dist.get_world_size()
            t = torch.tensor([data_time], dtype=torch.float64, device=vec.device)
            dist.all_reduce(t, op=dist.ReduceOp.MAX)       # the latency data loading causes is the slowest worker's
            dt = float(t.item()) if dist.get_rank() == 0 else data_time
            if dist.get_rank() != 0:
                return
        storage = get_event_storage()
        storage.put_scalar("data_time", dt)
        storage.put_scalar(f"{prefix}total_loss", vec.sum())
        if len(names) > 1:
            storage.put_scalars(**{k: vec[i] for i, k in enumerate(names)})

    def check_finite(self):
        """FloatingPointError like train_loop.py:L283-287, raised at the first host read after the offending step."""
        try:
            v = self.storage.history("total_loss").latest()
        except KeyError:
            return
        if not np.isfinite(v):
            raise FloatingPointError(f"Loss became infinite or NaN around iteration={self.iter}!\nlatest = {self.storage.latest()}")


class AMPTrainer(SimpleTrainer):
    """train_loop.py:L294-341 on the HIP path: needs a HipTrainer with amp=True (MODEL.COMPUTE_DTYPE fp16 + SOLVER.AMP)."""

    def __init__(self, model, data_loader, optimizer, grad_scaler=None):
        super().__init__(model, data_loader, optimizer)
        if grad_scaler is not None:
            raise TypeError("AMPTrainer: loss scaling lives inside the HipTrainer (device-side scale / found_inf / growth tracker); no GradScaler object")
        if not getattr(optimizer, "amp", False):
            raise ValueError("AMPTrainer needs a HipTrainer built with amp=True (cfg.SOLVER.AMP with MODEL.COMPUTE_DTYPE fp16)")

    @property
    def loss_scale(self):
        """Current scale (one host read; for logging at writer time, never per step)."""
        return float(self.optimizer.scale_state[0])
