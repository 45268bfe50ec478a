"""Hooks of the hook-driven trainers (reference: detectron2/engine/hooks.py:L42-381).

CallbackHook, IterationTimer, PeriodicWriter, PeriodicCheckpointer, LRScheduler and EvalHook keep the reference's timing rules (which
iteration they fire on, what they put into the storage).  AutogradProfiler (torch.autograd.profiler + chrome traces, hooks.py:L257-318)
becomes RocprofHook: HIP-event timing of the selected steps written to JSON, roctx ranges around them so that
`rocprofv3 --marker-trace --kernel-trace -- python train.py ...` attributes kernels to steps, and hipProfilerStart/Stop around the window
for collection-gated runs."""
import datetime
import json
import logging
import os
import time

import torch
import torch.distributed as dist

from ..checkpoint import PeriodicCheckpointer as _PeriodicCheckpointer
from ..utils.events import EventWriter
from .train_loop import HookBase

__all__ = ["CallbackHook", "IterationTimer", "PeriodicWriter", "PeriodicCheckpointer", "LRScheduler", "EvalHook", "RocprofHook"]


class CallbackHook(HookBase):
    def __init__(self, *, before_train=None, after_train=None, before_step=None, after_step=None):
        self._before_train, self._after_train, self._before_step, self._after_step = before_train, after_train, before_step, after_step

    def before_train(self):
        if self._before_train:
            self._before_train(self.trainer)

    def after_train(self):
        if self._after_train:
            self._after_train(self.trainer)
        del self._before_train, self._after_train, self._before_step, self._after_step      # closures may hold the trainer (hooks.py:L62-66)

    def before_step(self):
        if self._before_step:
            self._before_step(self.trainer)

    def after_step(self):
        if self._after_step:
            self._after_step(self.trainer)


class IterationTimer(HookBase):
    """Host wall time between before_step and after_step as the scalar "time" (after `warmup_iter` iterations), summary at the end.
    The step only ENQUEUES device work, so this is the rate the host feeds the GPU at; over many steps it equals the device rate because the
    hipGraph replays queue up behind each other (RocprofHook measures device time proper)."""

    def __init__(self, warmup_iter=3):
        self._warmup_iter = warmup_iter
        self._start_time = time.perf_counter()
        self._step_start = 0.0
        self._total = 0.0

    def before_train(self):
        self._start_time = time.perf_counter()
        self._total = 0.0

    def after_train(self):
        logger = logging.getLogger(__name__)
        total_time = time.perf_counter() - self._start_time
        num_iter = self.trainer.iter + 1 - self.trainer.start_iter - self._warmup_iter
        if num_iter > 0 and self._total > 0:
            logger.info("Overall training speed: %d iterations in %s (%.4f s / it)", num_iter, str(datetime.timedelta(seconds=int(self._total))),
                        self._total / num_iter)
        logger.info("Total training time: %s (%s on hooks)", str(datetime.timedelta(seconds=int(total_time))),
                    str(datetime.timedelta(seconds=int(total_time - self._total))))

    def before_step(self):
        self._step_start = time.perf_counter()

    def after_step(self):
        sec = time.perf_counter() - self._step_start
        iter_done = self.trainer.iter - self.trainer.start_iter + 1
        if iter_done >= self._warmup_iter:
            self.trainer.storage.put_scalars(time=sec)
            self._total += sec
        else:
            self._start_time = time.perf_counter()
            self._total = 0.0


class PeriodicWriter(HookBase):
    """writer.write() every `period` iterations and after the last one (hooks.py:L149-180); the trainer's finiteness check rides here,
    where the loss tensors are read anyway."""

    def __init__(self, writers, period=20):
        for w in writers:
            assert isinstance(w, EventWriter), w
        self._writers, self._period = writers, period

    def after_step(self):
        if (self.trainer.iter + 1) % self._period == 0 or self.trainer.iter == self.trainer.max_iter - 1:
            if hasattr(self.trainer, "check_finite"):
                self.trainer.check_finite()
            for w in self._writers:
                w.write()

    def after_train(self):
        for w in self._writers:
            w.write()
            w.close()


class PeriodicCheckpointer(_PeriodicCheckpointer, HookBase):
    """checkpoint.PeriodicCheckpointer as a hook: every `period` iterations and at max_iter - 1 (hooks.py:L183-199)."""

    def before_train(self):
        self.max_iter = self.trainer.max_iter

    def after_step(self):
        self.step(self.trainer.iter)


class LRScheduler(HookBase):
    """Puts the learning rate into the storage and advances the schedule after every iteration (hooks.py:L202-254).
    `scheduler`: a callable iteration -> list of per-group learning rates (engine.trainer.poly_lr / multistep_lr make the two rules of the
    projects), applied through HipTrainer.set_lr -- host-side values that travel in the next optimizer launch's arguments."""

    def __init__(self, optimizer=None, scheduler=None):
        self._optimizer, self._scheduler = optimizer, scheduler

    def before_train(self):
        self._optimizer = self._optimizer or self.trainer.optimizer
        if self._scheduler is None:
            self._scheduler = getattr(self.trainer, "scheduler", None)
        if self._scheduler is None:
            raise ValueError("LRScheduler: no schedule given")
        # the group whose rate is reported: the one with most parameters (hooks.py:L234-249)
        sizes = [len(g.named_params) for g in self._optimizer.groups]
        self._best_param_group_id = sizes.index(max(sizes))
        self._optimizer.set_lr(self._scheduler(self.trainer.iter))

    def after_step(self):
        lr = self._optimizer.groups[self._best_param_group_id].lr
        self.trainer.storage.put_scalar("lr", lr, smoothing_hint=False)
        self._optimizer.set_lr(self._scheduler(self.trainer.iter + 1))


def _flatten(d, prefix=""):
    out = {}
    for k, v in d.items():
        if isinstance(v, dict):
            out.update(_flatten(v, prefix + str(k) + "/"))
        else:
            out[prefix + str(k)] = v
    return out


class EvalHook(HookBase):
    """eval_function() every `eval_period` iterations and after the last one; its nested dict of floats goes into the storage flattened
    with '/' (hooks.py:L321-378)."""

    def __init__(self, eval_period, eval_function):
        self._period, self._func = eval_period, eval_function

    def _do_eval(self):
        results = self._func()
        if results:
            assert isinstance(results, dict), f"Eval function must return a dict. Got {results} instead."
            flat = _flatten(results)
            for k, v in flat.items():
                try:
                    flat[k] = float(v)
                except Exception as e:
                    raise ValueError(f"[EvalHook] eval_function should return a nested dict of float. Got '{k}: {v}' instead.") from e
            self.trainer.storage.put_scalars(**flat, smoothing_hint=False)
        if dist.is_available() and dist.is_initialized() and dist.get_world_size() > 1:
            dist.barrier()          # evaluation takes different time among workers

    def after_step(self):
        next_iter = self.trainer.iter + 1
        if self._period > 0 and next_iter % self._period == 0:
            self._do_eval()

    def after_train(self):
        if self.trainer.iter + 1 >= self.trainer.max_iter:      # not after a failed training
            self._do_eval()
        del self._func


class RocprofHook(HookBase):
    """The role of AutogradProfiler (hooks.py:L257-318) on ROCm.  For every iteration `enable_predicate(trainer)` selects:
      * a roctx range "step <iter>" (rocprofv3 --marker-trace groups the kernels of the step under it) and hipProfilerStart / Stop around
        the selected window;
      * two HIP events around run_step on the trainer's stream; device milliseconds per selected step are appended to
        <output_dir>/rocprof_steps.json when the window closes (one synchronisation per window, none per step).
    Run the process itself under `rocprofv3 --kernel-trace --marker-trace --stats -d <dir> -- python3 train.py ...` for kernel-level data."""

    def __init__(self, enable_predicate, output_dir):
        self._enable_predicate, self._output_dir = enable_predicate, output_dir
        self._open, self._events, self._active = False, [], False
        self.records = []

    @staticmethod
    def _cuda():
        return torch.cuda.is_available()

    def before_step(self):
        self._active = bool(self._enable_predicate(self.trainer))
        if not self._active:
            if self._open:
                self._close_window()
            return
        if self._cuda():
            if not self._open:
                torch.cuda.profiler.start()
            torch.cuda.nvtx.range_push(f"step {self.trainer.iter}")          # roctxRangePush on ROCm
            e0 = torch.cuda.Event(enable_timing=True)
            e0.record()
            self._events.append([self.trainer.iter, e0, None, time.perf_counter()])
        else:
            self._events.append([self.trainer.iter, None, None, time.perf_counter()])
        self._open = True

    def after_step(self):
        if not self._active:
            return
        ev = self._events[-1]
        if self._cuda():
            ev[2] = torch.cuda.Event(enable_timing=True)
            ev[2].record()
            torch.cuda.nvtx.range_pop()
        ev[3] = time.perf_counter() - ev[3]

    def _close_window(self):
        if self._cuda():
            torch.cuda.synchronize()
            torch.cuda.profiler.stop()
        for it, e0, e1, host in self._events:
            self.records.append({"iteration": it, "host_ms": host * 1e3, "device_ms": e0.elapsed_time(e1) if e0 is not None and e1 is not None else None})
        self._events, self._open = [], False
        os.makedirs(self._output_dir, exist_ok=True)
        with open(os.path.join(self._output_dir, "rocprof_steps.json"), "w") as f:
            json.dump(self.records, f)

    def after_train(self):
        if self._open:
            self._close_window()
