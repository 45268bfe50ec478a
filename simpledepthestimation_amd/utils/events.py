"""Event storage and writers of the hook-driven trainers (reference: detectron2/utils/events.py:L28-50 get_event_storage / EventWriter,
L52-131 JSONWriter, L184-269 CommonMetricPrinter, L272-520 EventStorage).

Same surface (put_scalar(s), history, latest, latest_with_smoothing_hint, step, iter, context manager, name_scope, write / close), one
deliberate difference: a scalar may be a 0-d DEVICE tensor.  It is kept as it is (cloned: under hipGraph replay loss tensors are static
buffers) and only turned into a float when a writer reads the storage, so a training step never forces a device -> host sync
(the reference's SimpleTrainer._write_metrics calls .cpu().item() on every loss every iteration, train_loop.py:L258)."""
import datetime
import json
import logging
import os
import time
from collections import defaultdict
from contextlib import contextmanager

import numpy as np
import torch

_CURRENT_STORAGE_STACK = []


def get_event_storage():
    assert len(_CURRENT_STORAGE_STACK), "get_event_storage() has to be called inside a 'with EventStorage(...)' context!"
    return _CURRENT_STORAGE_STACK[-1]


class _History:
    """(value, iteration) pairs of one scalar; device tensors are resolved lazily, all pending ones of a history with ONE copy."""

    def __init__(self):
        self._data = []

    def update(self, value, iteration):
        self._data.append((value, iteration))

    def _resolve(self):
        pend = [i for i, (v, _) in enumerate(self._data) if torch.is_tensor(v)]
        if pend:
            vals = torch.stack([self._data[i][0].reshape(()).float() for i in pend]).cpu().tolist()
            for i, x in zip(pend, vals):
                self._data[i] = (float(x), self._data[i][1])

    def values(self):
        self._resolve()
        return list(self._data)

    def latest(self):
        self._resolve()
        return self._data[-1][0]

    def median(self, window_size):
        self._resolve()
        return float(np.median([v for v, _ in self._data[-window_size:]]))

    def avg(self, window_size):
        self._resolve()
        return float(np.mean([v for v, _ in self._data[-window_size:]]))


class EventStorage:
    def __init__(self, start_iter=0):
        self._history = defaultdict(_History)
        self._smoothing_hints = {}
        self._latest_scalars = {}
        self._iter = start_iter
        self._current_prefix = ""

    def put_scalar(self, name, value, smoothing_hint=True):
        name = self._current_prefix + name
        if torch.is_tensor(value):
            value = value.detach().clone()
        else:
            value = float(value)
        self._history[name].update(value, self._iter)
        self._latest_scalars[name] = (value, self._iter)
        prev = self._smoothing_hints.get(name)
        if prev is not None:
            assert prev == smoothing_hint, f"Scalar {name} was put with a different smoothing_hint!"
        else:
            self._smoothing_hints[name] = smoothing_hint

    def put_scalars(self, *, smoothing_hint=True, **kwargs):
        for k, v in kwargs.items():
            self.put_scalar(k, v, smoothing_hint=smoothing_hint)

    def history(self, name):
        ret = self._history.get(name)
        if ret is None:
            raise KeyError(f"No history metric available for {name}!")
        return ret

    def histories(self):
        return self._history

    def latest(self):
        """{name: (value, iteration)} with every value a python float."""
        return {k: (self._history[k].latest(), it) for k, (_, it) in self._latest_scalars.items()}

    def latest_with_smoothing_hint(self, window_size=20):
        return {k: ((self._history[k].median(window_size) if self._smoothing_hints[k] else v), it) for k, (v, it) in self.latest().items()}

    def smoothing_hints(self):
        return self._smoothing_hints

    def step(self):
        self._iter += 1

    @property
    def iter(self):
        return self._iter

    @iter.setter
    def iter(self, val):
        self._iter = int(val)

    @property
    def iteration(self):
        return self._iter

    def __enter__(self):
        _CURRENT_STORAGE_STACK.append(self)
        return self

    def __exit__(self, exc_type, exc_val, exc_tb):
        assert _CURRENT_STORAGE_STACK[-1] is self
        _CURRENT_STORAGE_STACK.pop()

    @contextmanager
    def name_scope(self, name):
        old = self._current_prefix
        self._current_prefix = name.rstrip("/") + "/"
        yield
        self._current_prefix = old


class EventWriter:
    def write(self):
        raise NotImplementedError

    def close(self):
        pass


class JSONWriter(EventWriter):
    """One JSON object per line: {"iteration": i, name: smoothed value, ...} for every scalar updated since the last write."""

    def __init__(self, json_file, window_size=20):
        os.makedirs(os.path.dirname(os.path.abspath(json_file)), exist_ok=True)
        self._file_handle = open(json_file, "a")
        self._window_size = window_size
        self._last_write = -1

    def write(self):
        storage = get_event_storage()
        to_save = defaultdict(dict)
        for k, (v, it) in storage.latest_with_smoothing_hint(self._window_size).items():
            if it <= self._last_write:
                continue
            to_save[it][k] = v
        if to_save:
            self._last_write = max(to_save.keys())
        for it in sorted(to_save):
            rec = dict(to_save[it])
            rec["iteration"] = it
            self._file_handle.write(json.dumps(rec, sort_keys=True) + "\n")
        self._file_handle.flush()

    def close(self):
        self._file_handle.close()


class CommonMetricPrinter(EventWriter):
    """Console line with eta / iteration / losses / step and data time / lr / peak device memory (events.py:L184-269)."""

    def __init__(self, max_iter=None, window_size=20):
        self.logger = logging.getLogger(__name__)
        self._max_iter = max_iter
        self._window_size = window_size
        self._last_write = None
        self.last_line = None

    def _get_eta(self, storage):
        if self._max_iter is None:
            return ""
        it = storage.iter
        try:
            eta_seconds = storage.history("time").median(1000) * (self._max_iter - it - 1)
            storage.put_scalar("eta_seconds", eta_seconds, smoothing_hint=False)
            return str(datetime.timedelta(seconds=int(eta_seconds)))
        except KeyError:
            eta = None
            if self._last_write is not None:
                per_iter = (time.perf_counter() - self._last_write[1]) / max(1, it - self._last_write[0])
                eta = str(datetime.timedelta(seconds=int(per_iter * (self._max_iter - it - 1))))
            self._last_write = (it, time.perf_counter())
            return eta

    def write(self):
        storage = get_event_storage()
        it = storage.iter
        if it == self._max_iter:
            return          # the after_train call: nothing new (events.py:L229-233)
        def med(name, fn="median"):
            try:
                return getattr(storage.history(name), fn)(self._window_size if fn == "median" else 20)
            except KeyError:
                return None
        data_time, iter_time, lr = med("data_time", "avg"), med("time"), None
        try:
            lr = "{:.5g}".format(storage.history("lr").latest())
        except KeyError:
            lr = "N/A"
        eta = self._get_eta(storage)
        mem = torch.cuda.max_memory_allocated() / 1024.0 / 1024.0 if torch.cuda.is_available() else None
        losses = "  ".join("{}: {:.4g}".format(k, v.median(self._window_size)) for k, v in storage.histories().items() if "loss" in k)
        self.last_line = " {eta}iter: {it}  {losses}  {time}{data}lr: {lr}  {mem}".format(
            eta=f"eta: {eta}  " if eta else "", it=it, losses=losses,
            time="time: {:.4f}  ".format(iter_time) if iter_time is not None else "",
            data="data_time: {:.4f}  ".format(data_time) if data_time is not None else "", lr=lr,
            mem="max_mem: {:.0f}M".format(mem) if mem is not None else "")
        self.logger.info(self.last_line)
