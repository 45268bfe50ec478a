"""Checkpoint save / load / resume in the reference's on-disk format.

The reference's ``DetectionCheckpointer`` (detectron2/checkpoint/checkpoint.py:L9-45) is a thin subclass of fvcore's ``Checkpointer`` and its
``PeriodicCheckpointer`` is fvcore's; fvcore is a third-party dependency that is not under /root/reference and not installed here, so this
module restates the published behaviour the reference relies on at its call sites (projects/Supervised/train.py:L83-88,L134,
projects/MonoDepth2/train.py:L64-68,L111, utils/setup.py:L34):

* a checkpoint is ONE ``torch.save`` file ``<save_dir>/<name>.pth`` holding ``{"model": state_dict, <checkpointable name>: its state_dict ...,
  **extra}`` (e.g. ``"optimizer"``, ``"scheduler"``, ``"iteration"``); a bare state dict without a "model" key is also accepted on load (L24-28);
* ``<save_dir>/last_checkpoint`` is a text file naming the newest file; ``resume_or_load(path, resume=True)`` prefers it over `path`, and
  restores the checkpointables only when it resumes;
* on load a leading ``module.`` (DistributedDataParallel) is stripped when every key has it, tensors whose shape differs from the model's are
  dropped and reported, the rest is loaded non-strictly; missing ``pixel_mean`` / ``pixel_std`` buffers are not reported (L33-44);
* ``PeriodicCheckpointer.step(i)`` writes ``model_{i:07d}.pth`` every `period` steps and ``model_final.pth`` at ``max_iter - 1``, storing
  ``iteration=i`` in the file, and keeps at most `max_to_keep` periodic files.

The state-dict keys of the HIP modules are the reference's (fp32 master weights in the reference's [Cout, Cin, KH, KW] layout: the bf16 NHWC
operand copies are rebuilt by the batched pack kernel each step), so files written by either side load on the other.  The ResNet encoder keeps
torchvision's unused ``fc`` as an untrained module for the same reason; the trainer leaves it out of its parameter groups, and an optimizer
state written by the reference (whose encoder group ends with fc.weight / fc.bias) is matched group by group, position by position.

Files are read with ``torch.load(weights_only=True)`` only: nothing from a checkpoint is executed."""
import logging
import os
from collections import namedtuple

import torch
import torch.distributed as dist

Incompatible = namedtuple("Incompatible", ["missing_keys", "unexpected_keys", "incorrect_shapes"])
_LAST = "last_checkpoint"


def _is_main():
    return not (dist.is_available() and dist.is_initialized()) or dist.get_rank() == 0


def _cpu_tree(obj):
    """state dicts with every tensor detached on the host (views of the trainer's flat buffers become ordinary tensors)."""
    if torch.is_tensor(obj):
        return obj.detach().cpu().clone()
    if isinstance(obj, dict):
        return {k: _cpu_tree(v) for k, v in obj.items()}
    if isinstance(obj, (list, tuple)):
        return type(obj)(_cpu_tree(v) for v in obj)
    return obj


class DetectionCheckpointer:
    def __init__(self, model, save_dir="", *, save_to_disk=None, **checkpointables):
        self.model = model.module if hasattr(model, "module") and isinstance(getattr(model, "module"), torch.nn.Module) else model
        self.save_dir = save_dir
        self.save_to_disk = _is_main() if save_to_disk is None else bool(save_to_disk)
        self.checkpointables = dict(checkpointables)
        self.logger = logging.getLogger(__name__)

    # ---- writing -------------------------------------------------------------------------------------------------
    def save(self, name, **extra):
        if not self.save_dir or not self.save_to_disk:
            return None
        payload = {"model": _cpu_tree(self.model.state_dict())}
        for key, obj in self.checkpointables.items():
            payload[key] = _cpu_tree(obj.state_dict())
        payload.update(extra)
        os.makedirs(self.save_dir, exist_ok=True)
        fname = f"{name}.pth"
        path = os.path.join(self.save_dir, fname)
        tmp = path + ".tmp"
        torch.save(payload, tmp)
        os.replace(tmp, path)                      # a crash mid-write never leaves a truncated file under the final name
        self.tag_last_checkpoint(fname)
        return path

    def tag_last_checkpoint(self, basename):
        with open(os.path.join(self.save_dir, _LAST), "w") as f:
            f.write(basename)

    # ---- reading -------------------------------------------------------------------------------------------------
    def has_checkpoint(self):
        return bool(self.save_dir) and os.path.exists(os.path.join(self.save_dir, _LAST))

    def get_checkpoint_file(self):
        try:
            with open(os.path.join(self.save_dir, _LAST)) as f:
                return os.path.join(self.save_dir, f.read().strip())
        except OSError:
            return ""

    def get_all_checkpoint_files(self):
        if not self.save_dir or not os.path.isdir(self.save_dir):
            return []
        return sorted(os.path.join(self.save_dir, f) for f in os.listdir(self.save_dir) if f.endswith(".pth"))

    def resume_or_load(self, path, *, resume=True):
        if resume and self.has_checkpoint():
            return self.load(self.get_checkpoint_file())
        return self.load(path, checkpointables=[])

    def load(self, path, checkpointables=None):
        if not path:
            self.logger.info("No checkpoint found. Initializing model from scratch")
            return {}
        if not os.path.isfile(path):
            raise FileNotFoundError(f"Checkpoint {path} not found!")
        ckpt = torch.load(path, map_location="cpu", weights_only=True)
        if "model" not in ckpt:
            ckpt = {"model": ckpt}
        self.last_incompatible = self._load_model(ckpt.pop("model"))
        wanted = self.checkpointables if checkpointables is None else {k: self.checkpointables[k] for k in checkpointables}
        for key, obj in wanted.items():
            if key in ckpt:
                obj.load_state_dict(ckpt.pop(key))
        return ckpt                                 # what is left: "iteration" / "epoch" / anything the writer added

    def _load_model(self, state):
        state = dict(state)
        if state and all(k.startswith("module.") for k in state):
            state = {k[len("module."):]: v for k, v in state.items()}
        own = self.model.state_dict()
        wrong = []
        for k in list(state):
            if k in own and tuple(own[k].shape) != tuple(state[k].shape):
                wrong.append((k, tuple(state[k].shape), tuple(own[k].shape)))
                del state[k]
        res = self.model.load_state_dict(state, strict=False)
        missing = [k for k in res.missing_keys if k not in ("pixel_mean", "pixel_std")]
        out = Incompatible(missing, list(res.unexpected_keys), wrong)
        if missing:
            self.logger.warning("keys missing from the checkpoint: %s", ", ".join(missing))
        if out.unexpected_keys:
            self.logger.warning("checkpoint keys the model does not have: %s", ", ".join(out.unexpected_keys))
        for k, a, b in wrong:
            self.logger.warning("skipped '%s': checkpoint shape %s, model shape %s", k, a, b)
        return out


class PeriodicCheckpointer:
    def __init__(self, checkpointer, period, max_iter=None, max_to_keep=None, file_prefix="model"):
        self.checkpointer, self.period, self.max_iter = checkpointer, int(period), max_iter
        if max_to_keep is not None and max_to_keep <= 0:
            raise ValueError("max_to_keep must be positive")
        self.max_to_keep, self.file_prefix = max_to_keep, file_prefix
        self.recent = []

    def step(self, iteration, **extra):
        iteration = int(iteration)
        extra = dict(extra, iteration=iteration)
        if (iteration + 1) % self.period == 0:
            path = self.checkpointer.save(f"{self.file_prefix}_{iteration:07d}", **extra)
            if path is not None and self.max_to_keep is not None:
                self.recent.append(path)
                while len(self.recent) > self.max_to_keep:
                    old = self.recent.pop(0)
                    if os.path.exists(old) and not old.endswith(f"{self.file_prefix}_final.pth"):
                        os.remove(old)
        if self.max_iter is not None and iteration >= self.max_iter - 1:
            self.checkpointer.save(f"{self.file_prefix}_final", **extra)

    def save(self, name, **extra):
        return self.checkpointer.save(name, **extra)
