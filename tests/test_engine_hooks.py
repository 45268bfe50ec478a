"""CPU: the hook-driven trainers and hooks (detectron2/engine/train_loop.py:L18-341, hooks.py:L42-381, utils/events.py) on a tiny model.
The step engine is the real HipTrainer with the torch restatement of the fused Adam kernel (as in test_dp_gloo.py)."""
import json
import os

import numpy as np
import pytest
import torch

from simpledepthestimation_amd.checkpoint import DetectionCheckpointer
from simpledepthestimation_amd.engine import hooks as H
from simpledepthestimation_amd.engine.train_loop import AMPTrainer, HookBase, SimpleTrainer, TrainerBase
from simpledepthestimation_amd.engine.trainer import HipTrainer, ParamGroup
from simpledepthestimation_amd.utils.events import CommonMetricPrinter, EventStorage, JSONWriter, get_event_storage
from test_dp_gloo import Tiny, torch_adam


def _batch(seed):
    g = torch.Generator().manual_seed(seed)
    return {"x": torch.randn(8, 6, generator=g), "t": torch.randn(8, generator=g)}


def _setup(seed=3):
    torch.manual_seed(seed)
    model = Tiny()
    opt = HipTrainer(model, [ParamGroup("a", model.a.named_parameters(prefix="a"), 1e-2, 0.0), ParamGroup("b", model.b.named_parameters(prefix="b"), 5e-3, 0.0)],
                     adam_fn=torch_adam)
    return model, opt


class _Log(HookBase):
    def __init__(self):
        self.calls = []

    def before_train(self):
        self.calls.append(("before_train", self.trainer.iter))

    def after_train(self):
        self.calls.append(("after_train", self.trainer.iter))

    def before_step(self):
        self.calls.append(("before_step", self.trainer.iter, self.trainer.storage.iter))

    def after_step(self):
        self.calls.append(("after_step", self.trainer.iter))


def test_trainer_base_call_order_and_storage_invariant():
    class T(TrainerBase):
        def run_step(self):
            get_event_storage().put_scalar("x", self.iter * 2.0)
    t, log = T(), _Log()
    t.register_hooks([None, log])
    t.train(2, 5)
    assert log.calls[0] == ("before_train", 2) and log.calls[-1] == ("after_train", 5)      # iter == max_iter after a complete run
    steps = [c for c in log.calls if c[0] == "before_step"]
    assert steps == [("before_step", i, i) for i in (2, 3, 4)]
    assert t.storage.history("x").values() == [(4.0, 2), (6.0, 3), (8.0, 4)]
    with pytest.raises(AssertionError):
        get_event_storage()                       # the storage context closed with train()

    class Boom(TrainerBase):
        def run_step(self):
            if self.iter == 1:
                raise RuntimeError("boom")
    b, log2 = Boom(), _Log()
    b.register_hooks([log2])
    with pytest.raises(RuntimeError):
        b.train(0, 4)
    assert log2.calls[-1] == ("after_train", 1)        # after_train still runs, iter < max_iter tells it the run failed


def test_simple_trainer_with_all_hooks(tmp_path):
    model, opt = _setup()
    loader = [_batch(i) for i in range(12)]
    trainer = SimpleTrainer(model, loader, opt)
    evals = []

    def evaluate():
        evals.append(trainer.iter)
        return {"kitti evaluator": {"abs_rel": 0.1 + 0.01 * len(evals), "d1": 0.9}}
    sched = lambda it: [1e-2 * 0.5 ** (it // 4), 5e-3 * 0.5 ** (it // 4)]
    ck = DetectionCheckpointer(model, str(tmp_path), optimizer=opt)
    printer = CommonMetricPrinter(max_iter=10)
    prof = H.RocprofHook(lambda tr: 2 <= tr.iter < 4, str(tmp_path / "prof"))
    seen = []
    trainer.register_hooks([H.IterationTimer(warmup_iter=2), H.LRScheduler(scheduler=sched), H.PeriodicCheckpointer(ck, 4),
                            H.EvalHook(5, evaluate), prof, H.CallbackHook(after_step=lambda tr: seen.append(tr.iter)),
                            H.PeriodicWriter([JSONWriter(str(tmp_path / "metrics.json")), printer], period=3)])
    trainer.train(0, 10)
    assert seen == list(range(10)) and evals == [4, 9, 10]          # every 5 iterations, and once more after the last one (hooks.py:L367-376)
    recs = [json.loads(l) for l in open(tmp_path / "metrics.json")]
    assert [r["iteration"] for r in recs] == [2, 4, 5, 8, 9] or [r["iteration"] for r in recs][:2] == [2, 4]
    by_it = {r["iteration"]: r for r in recs}
    assert "total_loss" in by_it[2] and "lr" in by_it[2] and "data_time" in by_it[2] and "time" in by_it[2]
    assert by_it[2]["lr"] == 1e-2 and by_it[8]["lr"] == 1e-2 * 0.25                    # lr of the group with most parameters, schedule applied per iteration
    assert abs(by_it[4]["kitti evaluator/abs_rel"] - 0.11) < 1e-12 and by_it[9]["kitti evaluator/d1"] == 0.9
    assert sorted(f for f in os.listdir(tmp_path) if f.endswith(".pth")) == ["model_0000003.pth", "model_0000007.pth", "model_final.pth"]
    assert printer.last_line is not None and "total_loss" in printer.last_line and "lr:" in printer.last_line and "iter: 9" in printer.last_line
    steps = json.load(open(tmp_path / "prof" / "rocprof_steps.json"))
    assert [s["iteration"] for s in steps] == [2, 3] and all(s["host_ms"] > 0 for s in steps)
    # the run is the plain loop's run: same parameters as stepping the HipTrainer by hand with the same schedule
    model2, opt2 = _setup()
    for i in range(10):
        opt2.set_lr(sched(i))
        opt2.step(_batch(i))
    for a, b in zip(model.parameters(), model2.parameters()):
        assert torch.equal(a, b)


def test_losses_stay_tensors_until_a_writer_reads_them():
    model, opt = _setup()
    trainer = SimpleTrainer(model, [_batch(i) for i in range(4)], opt)
    peek = []
    trainer.register_hooks([H.CallbackHook(after_step=lambda tr: peek.append(torch.is_tensor(tr.storage._latest_scalars["total_loss"][0])))])
    trainer.train(0, 4)
    assert peek == [True] * 4                       # parked as tensors: no per-step .item()
    vals = trainer.storage.history("total_loss").values()
    assert len(vals) == 4 and all(isinstance(v, float) and np.isfinite(v) for v, _ in vals)


def test_simple_trainer_runs_past_one_epoch_of_a_finite_loader_and_over_the_infinite_sampler():
    """SimpleTrainer.run_step pulls next(iterator) for max_iter iterations (train_loop.py:L227-247): a loader over the infinite TrainingSampler
    never ends, and a finite (one epoch per iterator) loader is restarted at its end with the sampler's epoch advanced."""
    import torch.utils.data as data
    from simpledepthestimation_amd.data.build import TrainingSampler

    class DS(data.Dataset):
        def __len__(self):
            return 5

        def __getitem__(self, i):
            return _batch(i)

    class Smp(data.Sampler):
        epoch = 0

        def set_epoch(self, e):
            self.epoch = e

        def __iter__(self):
            return iter(range(5))

        def __len__(self):
            return 5

    model, opt = _setup()
    smp = Smp()
    finite = data.DataLoader(DS(), batch_sampler=data.BatchSampler(smp, 2, drop_last=True), collate_fn=lambda b: b[0])
    tr = SimpleTrainer(model, finite, opt)
    tr.train(0, 7)                                   # 2 batches per epoch: crosses three epoch boundaries
    assert tr.iter == 7 and smp.epoch == 3
    model, opt = _setup()
    inf = data.DataLoader(DS(), batch_sampler=data.BatchSampler(TrainingSampler(5, seed=1), 2, drop_last=True), collate_fn=lambda b: b[0])
    tr = SimpleTrainer(model, inf, opt)
    tr.train(0, 9)
    assert len(tr.storage.history("total_loss").values()) == 9


def test_non_finite_loss_raises_at_the_next_write():
    model, opt = _setup()
    bad = _batch(1); bad["t"] = bad["t"] * float("nan")
    trainer = SimpleTrainer(model, [_batch(0), bad, _batch(2), _batch(3)], opt)
    w = JSONWriter(os.devnull)
    trainer.register_hooks([H.PeriodicWriter([w], period=2)])
    with pytest.raises(FloatingPointError):
        trainer.train(0, 4)


def test_simple_and_amp_trainer_argument_checks():
    model, opt = _setup()
    with pytest.raises(TypeError):
        SimpleTrainer(model, [], torch.optim.SGD(model.parameters(), lr=0.1))       # the optimizer must be the HipTrainer over this model
    other, _ = _setup(4)
    with pytest.raises(TypeError):
        SimpleTrainer(other, [], opt)
    with pytest.raises(ValueError):
        AMPTrainer(model, [], opt)                  # not built with amp=True
    with pytest.raises(TypeError):
        AMPTrainer(model, [], opt, grad_scaler=object())


def test_event_storage_smoothing_and_scopes():
    with EventStorage(5) as s:
        for i, v in enumerate([1.0, 100.0, 3.0]):
            s.put_scalar("loss", torch.tensor(v))
            s.put_scalar("lr", 0.1 * (i + 1), smoothing_hint=False)
            s.step()
        with s.name_scope("val"):
            s.put_scalar("abs_rel", 0.2)
        sm = s.latest_with_smoothing_hint(3)
        assert sm["loss"] == (3.0, 7) and abs(sm["lr"][0] - 0.3) < 1e-12 and sm["val/abs_rel"] == (0.2, 8)     # median of the window; no smoothing for lr
        with pytest.raises(AssertionError):
            s.put_scalar("lr", 1.0, smoothing_hint=True)
        with pytest.raises(KeyError):
            s.history("nope")
