"""Evaluator plumbing with the reference's surface (detectron2/evaluation/evaluator.py:L16-217): EVALUATOR_REGISTRY, build_evaluator(cfg, dir),
DatasetEvaluator {reset, process(inputs, outputs), evaluate()}, DatasetEvaluators, inference_context, inference_on_dataset.

Difference by design: ``process`` leaves the per-image metrics on the GPU (no ``to_numpy`` of predictions, no per-batch synchronisation);
the one device->host copy of the whole run happens in ``evaluate()``."""
import logging
import time
from collections import OrderedDict
from contextlib import contextmanager

import torch
import torch.distributed as dist

from ..utils.registry import Registry

EVALUATOR_REGISTRY = Registry("EVALUATOR")


def _rank0():
    return not (dist.is_available() and dist.is_initialized()) or dist.get_rank() == 0


def build_evaluator(cfg, ouput_folder):
    built = [EVALUATOR_REGISTRY.get(name)(cfg, ouput_folder) for name in cfg.EVALUATORS]
    for ev in built:
        assert isinstance(ev, DatasetEvaluator), type(ev)
    return built


class DatasetEvaluator:
    """Accumulates over ``process`` calls, summarises in ``evaluate`` (returns {task: {metric: value}} on rank 0, {} elsewhere)."""

    def __init__(self, cfg=None):
        # names of the test-time preprocess steps, forward order; their backward() chain is folded into index maps by the evaluators
        steps = cfg.DATASETS.TEST.get("PREPROCESS", []) if (cfg is not None and "DATASETS" in cfg and "TEST" in cfg.DATASETS) else []
        names = [s["NAME"] if isinstance(s, dict) else s.NAME for s in steps]
        self.preprocess_chain = [n for n in names if n in ("Resize", "KBCrop", "CropTopTo")]      # the steps that have a backward() (augmentation.py:L67,L113,L163)

    def reset(self):
        pass

    def process(self, inputs, outputs):
        pass

    def evaluate(self):
        pass


class DatasetEvaluators(DatasetEvaluator):
    def __init__(self, evaluators):
        super().__init__()
        self._evaluators = list(evaluators)

    def reset(self):
        for ev in self._evaluators:
            ev.reset()

    def process(self, inputs, outputs):
        for ev in self._evaluators:
            ev.process(inputs, outputs)

    def evaluate(self):
        merged = OrderedDict()
        for ev in self._evaluators:
            res = ev.evaluate()
            if _rank0() and res is not None:
                for task, table in res.items():
                    assert task not in merged, "Different evaluators produce results with the same key {}".format(task)
                    merged[task] = table
        return merged


@contextmanager
def inference_context(model):
    was_training = model.training
    model.eval()
    try:
        yield
    finally:
        model.train(was_training)


def inference_on_dataset(model, data_loader, evaluator):
    """Runs `model` (eval mode, no grad) over `data_loader`, feeds `evaluator`, returns ``evaluator.evaluate()`` ({} if None).
    `evaluator` may be one DatasetEvaluator, a list of them (what build_evaluator returns) or None.  Logs seconds per batch measured
    with ONE synchronisation at the end of the run instead of one per batch."""
    log = logging.getLogger(__name__)
    if evaluator is None:
        evaluator = DatasetEvaluators([])
    elif isinstance(evaluator, (list, tuple)):
        evaluator = DatasetEvaluators(evaluator)
    evaluator.reset()
    n = 0
    was_training = getattr(model, "training", False)
    if isinstance(model, torch.nn.Module):
        model.eval()
    t0 = time.perf_counter()
    try:
        with torch.no_grad():
            for inputs in data_loader:
                evaluator.process(inputs, model(inputs))
                n += 1
    finally:
        if isinstance(model, torch.nn.Module):
            model.train(was_training)
    if torch.cuda.is_available():
        torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    log.info("Total inference time: %.3f s (%.6f s / batch, %d batches)", dt, dt / max(n, 1), n)
    res = evaluator.evaluate()
    return {} if res is None else res
