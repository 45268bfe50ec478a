"""The training loops of the two projects on the HIP path (the part the reference actually runs:
projects/MonoDepth2/train.py:L44-121 ``do_train`` / L35-41 ``do_test`` and projects/Supervised/train.py:L70-140).

Same order of events as the reference: resume_or_load -> for epoch: for batch: step, log every LOG_PERIOD -> LR schedule (MultiStepLR per
epoch / polynomial per iteration) -> periodic checkpoint -> evaluation every TEST.EVAL_PERIOD epochs.  What differs, on purpose:

* the step is ``HipTrainer.step`` (graph replay, fused Adam, RCCL all-reduce inside) instead of model / backward / optimizer.step;
* the reference calls ``.item()`` on every loss every iteration (a host sync per step, train.py:L95); here the 0-d loss tensors are added
  into a device-side accumulator and read back once per LOG_PERIOD -- the finiteness assert (L93) moves to that read-back;
* scalars go to ``OUTPUT_DIR/metrics.json`` as JSON lines (the JSONWriter format of utils/events.py:L52-131: one object per write with
  "iteration" and the scalars); TensorBoard / the console printer are not rebuilt.

The data loaders are the caller's (any iterable with ``len`` yielding the reference's batch dicts)."""
import json
import logging
import os

import torch
import torch.distributed as dist

from ..checkpoint import DetectionCheckpointer, PeriodicCheckpointer
from ..evaluation import build_evaluator, inference_on_dataset
from . import trainer as T

log = logging.getLogger(__name__)


def _main():
    return not (dist.is_available() and dist.is_initialized()) or dist.get_rank() == 0


class EpochSchedule:
    """The scheduler checkpointable: the epoch / iteration counters the LR rules are functions of."""

    def __init__(self):
        self.last_epoch, self.global_step = 0, 0

    def state_dict(self):
        return {"last_epoch": self.last_epoch, "global_step": self.global_step}

    def load_state_dict(self, sd):
        self.last_epoch, self.global_step = int(sd.get("last_epoch", 0)), int(sd.get("global_step", 0))


class _LossMeter:
    """Device-side running sums of the step's loss scalars; one host read per flush."""

    def __init__(self):
        self.names, self.acc, self.n = None, None, 0

    def add(self, loss_dict):
        if self.names is None:
            self.names = sorted(loss_dict)
        vec = torch.stack([loss_dict[k].detach().float() for k in self.names])
        self.acc = vec.clone() if self.acc is None else self.acc.add_(vec)        # clone: under graph replay the loss tensors are static buffers
        self.n += 1

    def flush(self):
        if not self.n:
            return {}
        acc = self.acc / self.n
        if dist.is_available() and dist.is_initialized() and dist.get_world_size() > 1:
            dist.all_reduce(acc)                                                  # comm.reduce_dict(average=True), train.py:L95
            acc = acc / dist.get_world_size()
        vals = acc.cpu().tolist()
        self.acc, self.n = None, 0
        out = dict(zip(self.names, vals))
        assert all(v == v and abs(v) != float("inf") for v in vals), out       # train.py:L93
        return out


def do_test(cfg, model, data_loader_test):
    # projects/*/train.py:L62-66: OUTPUT_DIR/inference/<DATASETS.TEST.NAME>
    name = cfg.DATASETS.TEST.get("NAME", "") if ("DATASETS" in cfg and "TEST" in cfg.DATASETS) else ""
    return inference_on_dataset(model, data_loader_test, build_evaluator(cfg, os.path.join(cfg.OUTPUT_DIR, "inference", name) if name else
                                                                         os.path.join(cfg.OUTPUT_DIR, "inference")))


def do_train(cfg, model, data_loader, data_loader_test=None, resume=False, use_graph=None):
    """Returns the list of records written to metrics.json.  `model` may be wrapped (FakeDDP) or bare."""
    core = model.module if hasattr(model, "module") and isinstance(model.module, torch.nn.Module) else model
    core.train()
    supervised = cfg.MODEL.META_ARCHITECTURE == "SupDepthModel"
    graph = (next(core.parameters()).is_cuda if use_graph is None else use_graph)
    tr = (T.supervised_trainer if supervised else T.monodepth2_trainer)(core, cfg, use_graph=graph)
    sched = EpochSchedule()
    ckpt = DetectionCheckpointer(core, cfg.OUTPUT_DIR, optimizer=tr, scheduler=sched)
    periodic = PeriodicCheckpointer(ckpt, cfg.SOLVER.CHECKPOINT_PERIOD, max_iter=cfg.SOLVER.MAX_EPOCHS)
    # MonoDepth2/train.py:L68 reads "iteration" (what PeriodicCheckpointer writes); Supervised/train.py:L88 reads "epoch", which is never
    # written, so the reference's supervised loop always restarts at epoch 0 -- here both resume from the stored epoch
    start_epoch = ckpt.resume_or_load(cfg.MODEL.WEIGHTS, resume=resume).get("iteration", -1) + 1
    per_epoch = len(data_loader)
    max_iter = cfg.SOLVER.MAX_EPOCHS * per_epoch
    global_step = start_epoch * per_epoch
    base = [cfg.SOLVER.DEPTH_LR, cfg.SOLVER.POSE_LR]        # MonoDepth2's two groups (Depth, Pose); the supervised rule below has one value

    def epoch_lrs(epoch):
        return [T.multistep_lr(b, epoch, cfg.SOLVER.LR_STEPS, cfg.SOLVER.GAMMA) for b in base]

    if not supervised:
        tr.set_lr(epoch_lrs(start_epoch))
    meter, records = _LossMeter(), []
    out_path = os.path.join(cfg.OUTPUT_DIR, "metrics.json")
    if _main():
        os.makedirs(cfg.OUTPUT_DIR, exist_ok=True)

    def write(extra=None):
        rec = {"iteration": global_step, "epoch": epoch, "lr": tr.groups[0].lr}
        losses = meter.flush()
        if losses:
            rec.update(total_loss=sum(losses.values()), **losses)
        rec.update(extra or {})
        if _main():
            with open(out_path, "a") as f:
                f.write(json.dumps(rec, sort_keys=True) + "\n")
            log.info("epoch %d iter %d: %s", epoch, global_step, {k: (round(v, 5) if isinstance(v, float) else v) for k, v in rec.items()})
        records.append(rec)

    log.info("Starting training from iteration %d", start_epoch)
    for epoch in range(start_epoch, cfg.SOLVER.MAX_EPOCHS):
        for epoch_iter, data in enumerate(data_loader):
            global_step += 1
            if supervised:      # polynomial decay (Supervised/train.py:L125-128): the reference sets it AFTER step g from g, so step g runs on f(g - 1)
                tr.set_lr([T.poly_lr(cfg, global_step - 1, max_iter)] * len(tr.groups))
            meter.add(tr.step(data))
            if (epoch_iter + 1) % cfg.LOG_PERIOD == 0:
                write()
        if not supervised:
            tr.set_lr(epoch_lrs(epoch + 1))                      # scheduler.step() at the end of the epoch (MonoDepth2/train.py:L109)
        sched.last_epoch, sched.global_step = epoch + 1, global_step
        periodic.step(epoch)
        if cfg.TEST.EVAL_PERIOD > 0 and (epoch + 1) % cfg.TEST.EVAL_PERIOD == 0 and data_loader_test is not None:
            results = do_test(cfg, core, data_loader_test)
            core.train()
            write({f"{tag}/{k}": v for tag, table in results.items() for k, v in table.items()})
    return records
