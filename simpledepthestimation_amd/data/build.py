"""Dataset registry and data loaders (reference: detectron2/data/build.py:L35-140, samplers/distributed_sampler.py:L57-84).

``DatasetBase`` runs the configured preprocess chain forward (and backward on predictions); the train loader shards with
torch's DistributedSampler (per-rank batch = IMS_PER_BATCH / world, drop_last), the test loader walks contiguous per-rank shards one image
at a time.  ``DevicePrefetcher`` is the MI355X-side addition: batches are staged in pinned host memory and copied to HBM on a copy stream
while the previous step's hipGraph replays, so the step never waits on PCIe (35-106 MB of fp32 images per step)."""
import logging

import ctypes

import numpy as np
import torch
import torch.distributed as dist
import torch.utils.data as data

from ..utils.registry import Registry
from .preprocess import build_preprocess

DATASET_REGISTRY = Registry("DATASET")


def _world():
    return (dist.get_world_size(), dist.get_rank()) if (dist.is_available() and dist.is_initialized()) else (1, 0)


class DatasetBase(data.Dataset):
    def __init__(self, dataset_cfg, cfg):
        self.preprocesses = [build_preprocess(p) for p in dataset_cfg.get("PREPROCESS", [])]

    def __getitem__(self, item):
        raise NotImplementedError

    def preprocess(self, data_dict):
        for p in self.preprocesses:
            data_dict = p.forward(data_dict)
        return data_dict

    def get_prediction(self, data_dict):
        for p in self.preprocesses[::-1]:
            data_dict = p.backward(data_dict)
        return data_dict

    def batch_collator(self, batch):
        return data.default_collate(batch)


class InferenceSampler(data.Sampler):
    """Contiguous shards of exactly the dataset (ranks may differ by one sample), distributed_sampler.py:L57-84."""

    def __init__(self, size):
        assert size > 0
        world, rank = _world()
        shard = (size - 1) // world + 1
        self._local_indices = range(shard * rank, min(shard * (rank + 1), size))

    def __iter__(self):
        yield from self._local_indices

    def __len__(self):
        return len(self._local_indices)


class TrainingSampler(data.Sampler):
    """The reference's infinite training stream (samplers/distributed_sampler.py:L12-52): the concatenation of seeded permutations of
    range(size) (or of range(size) itself without shuffling), of which rank r of w takes every w-th index starting at r.  The hook-driven
    trainers (engine/train_loop.py: SimpleTrainer.run_step) pull `next(iterator)` for max_iter iterations and rely on it never ending.
    seed=None: one seed for all ranks, drawn by rank 0 and broadcast (the role of comm.shared_random_seed)."""

    def __init__(self, size, shuffle=True, seed=None):
        if size <= 0:
            raise ValueError("TrainingSampler needs a non-empty dataset")
        self._size, self._shuffle = int(size), bool(shuffle)
        self._world, self._rank = _world()
        if seed is None:
            seed = int(np.random.randint(2 ** 31))
            if self._world > 1:
                box = [seed]
                dist.broadcast_object_list(box, src=0)
                seed = int(box[0])
        self._seed = int(seed)

    def __iter__(self):
        g = torch.Generator()
        g.manual_seed(self._seed)
        pos = self._rank                     # position of this rank's next index inside the current permutation
        while True:
            order = torch.randperm(self._size, generator=g) if self._shuffle else torch.arange(self._size)
            while pos < self._size:
                yield int(order[pos])
                pos += self._world
            pos -= self._size                # the stride carries over the seam between two permutations


def _seed_worker(worker_id):
    import random
    seed = np.random.randint(2 ** 31) + worker_id
    np.random.seed(seed % (2 ** 32))
    torch.manual_seed(seed)
    random.seed(seed)


def build_batch_data_loader(dataset, sampler, total_batch_size, *, num_workers=0):
    world, _ = _world()
    assert total_batch_size > 0 and total_batch_size % world == 0, \
        f"Total batch size ({total_batch_size}) must be divisible by the number of gpus ({world})."
    batch_sampler = data.BatchSampler(sampler, total_batch_size // world, drop_last=True)      # static shapes: what the captured hipGraph needs, too
    return data.DataLoader(dataset, num_workers=num_workers, batch_sampler=batch_sampler, collate_fn=dataset.batch_collator, worker_init_fn=_seed_worker)


def build_detection_train_loader(cfg):
    dataset = DATASET_REGISTRY.get(cfg.DATASETS.TRAIN.NAME)(cfg.DATASETS.TRAIN, cfg)
    assert isinstance(dataset, DatasetBase)
    name = cfg.DATALOADER.get("SAMPLER_TRAIN", "DDPSampler")
    logging.getLogger(__name__).info("Using training sampler %s", name)
    world, rank = _world()
    if name == "TrainingSampler":            # build.py:L108-109: the infinite stream the hook-driven trainers iterate with a bare next()
        sampler = TrainingSampler(len(dataset))
    elif name == "DDPSampler":
        sampler = data.distributed.DistributedSampler(dataset, num_replicas=world, rank=rank, shuffle=True)
    else:
        raise ValueError(f"Unknown training sampler: {name}")
    return build_batch_data_loader(dataset, sampler, cfg.SOLVER.IMS_PER_BATCH, num_workers=cfg.DATALOADER.NUM_WORKERS)


def build_detection_test_loader(cfg):
    if "TEST" not in cfg.DATASETS:
        return None
    dataset = DATASET_REGISTRY.get(cfg.DATASETS.TEST.NAME)(cfg.DATASETS.TEST, cfg)
    assert isinstance(dataset, DatasetBase)
    batch_sampler = data.BatchSampler(InferenceSampler(len(dataset)), 1, drop_last=False)      # one image per step at inference
    return data.DataLoader(dataset, num_workers=cfg.DATALOADER.NUM_WORKERS, batch_sampler=batch_sampler, collate_fn=dataset.batch_collator)


def _mark(name):
    """Timeline marker on the current stream when a diagnostic run switched them on (hip.lib.marks_enable); a no-op otherwise."""
    from ..hip import lib as L
    if L.MARKS is not None:
        L.mark(name)


class DevicePrefetcher:
    """Wraps a loader: the next `ahead` batches are copied host -> device on a copy stream while step k runs.  Arrays (tensors, numpy arrays, lists of
    them) move; everything else (flip, metadata) passes through.  len() and re-iteration follow the wrapped loader.

    The device side is a ring of `slots` persistent buffer sets (static shapes: what the captured hipGraph needs, too): entry `k` of a batch lands in
    the same device tensor every `slots` batches, so the steady state allocates nothing -- per-step allocations on a side stream make the caching
    allocator wait for (or hipMalloc around) blocks the main stream still uses, and the host ends up pacing the GPU (measured: 7.6 ms of host time
    per 6.6 ms step).

    Hand-over without events: a one-thread kernel behind the uploads stores the batch's sequence number into pinned host memory (sde_store_u64), the
    host hands the batch over once it has SEEN that number (two batches ahead it normally has), and the consumer's stream needs no cross-stream wait;
    the other way round, a flag store enqueued on the consumer's stream when it comes back for the next batch tells the host that the slot may be
    rewritten.  The hipEventRecord / hipStreamWaitEvent pairs this replaces left the GPU idle for 0.5 ms per step between two replays of the step
    graph (MonoDepth2-R18: 4.68 ms/step with events, 4.07 with resident inputs; in-graph markers and a variant probe in profiles/README.md, round 3).
    A yielded batch is valid until the iterator is advanced `slots - ahead` more times.
    device_aug: optional callable(batch, buffers) run on the copy stream right behind the uploads (data/device_aug.py: DeviceImageAug turns the raw
    uint8 frames of the ON_DEVICE preprocess chain into the fp32 image entries).  Without it the raw entries are uploaded like any other array and
    the consumer transforms them -- HipTrainer.input_transform = DeviceImageAug(...) runs the kernels at the head of the step, straight into the
    captured graph's static inputs; that is the faster arrangement on this stack (bench.py --with-loader; DESIGN.md: measured both ways)."""

    _RAW = ("img_u8", "ctx_img_u8", "aug_params", "device_resize")     # consumed by device_aug (which uploads them itself)

    def __init__(self, loader, device, device_aug=None, slots=4, ahead=2, pick_stream=True):
        self.loader, self.device = loader, torch.device(device)
        self.device_aug = device_aug
        self._stream = torch.cuda.Stream(device=self.device) if self.device.type == "cuda" else None
        # Which copy stream: the HIP runtime multiplexes all streams onto four hardware queues, bound on first use, and an upload that shares the queue of
        # the training stream (or of a critical branch of its step graph) holds back what is submitted behind it.  Measured with 50 MB of uploads per step
        # (MonoDepth2-R18, resident inputs 4.05 ms/step): the stream created here, first used BEFORE the consumer captured its graph, 4.65; a stream first
        # used once the graph is replaying 4.22 (= resident + the resize kernels).  Supervised-R50 the other way round: 6.60 against 7.05; a high-priority
        # stream: MonoDepth2 4.22, Supervised 15 ms.  HIP offers no way to choose the queue, so the prefetcher tries both and keeps the faster one:
        # batches 5-16 on this stream, 21-32 on a fresh one, judged by the median hand-over interval (the per-slot flags make a switch safe at any time).
        # pick_stream=False keeps the first stream.  (profiles/r03x_loader_probe.txt)
        self._pick = {"phase": 0, "t": [], "med": {}} if (pick_stream and self._stream is not None) else None
        self._slots = max(2, int(slots))
        # batches staged ahead of the one being consumed.  Two, not one: HIP streams share a handful of hardware queues, and an upload that lands on
        # the queue of one of the step graph's branches starts only when that branch has drained, i.e. at the END of the running step -- one batch
        # ahead, the next step then began by waiting 0.5 ms for its frames (MonoDepth2-R18, in-graph markers); two ahead it has a whole step of slack
        self._ahead = max(1, min(int(ahead), self._slots - 1))
        # hand-over flags in pinned host memory, written in stream order by one-thread kernels (sde_store_u64) and polled by the host:
        # _flags[0][slot] = sequence number of the last upload that COMPLETED into the slot, _flags[1][slot] = of the last batch the consumer is DONE with
        self._flags = torch.zeros(2, self._slots, dtype=torch.int64).pin_memory() if self._stream is not None else None
        self._seq = [0] * self._slots                         # per slot: sequence number (1-based batch count) of the batch staged into it last
        self._bufs = [dict() for _ in range(self._slots)]      # per slot: key -> persistent device tensor

    def __len__(self):
        return len(self.loader)

    def _into(self, bufs, key, v):
        """Copy host / device tensor v into the slot's persistent buffer for `key` (re-created when the shape or dtype changes)."""
        if isinstance(v, np.ndarray):
            v = torch.from_numpy(np.ascontiguousarray(v))
        b = bufs.get(key)
        if b is None or b.shape != v.shape or b.dtype != v.dtype:
            b = bufs[key] = torch.empty(v.shape, dtype=v.dtype, device=self.device)
        if v.device.type == "cpu" and not v.is_pinned():
            v = v.pin_memory()
        b.copy_(v, non_blocking=True)
        return b

    def _move(self, bufs, key, v):
        if isinstance(v, np.ndarray) or torch.is_tensor(v):
            return self._into(bufs, key, v) if self._stream is not None else torch.as_tensor(v).to(self.device)
        if isinstance(v, (list, tuple)) and v and all(isinstance(x, np.ndarray) or torch.is_tensor(x) for x in v):
            return [self._move(bufs, (key, i), x) for i, x in enumerate(v)]
        return v

    def _stage(self, batch, slot):
        if self._stream is None:
            out = {k: self._move(None, k, v) for k, v in batch.items()}
            return (self.device_aug(out) if self.device_aug is not None else out), None
        bufs = self._bufs[slot]
        self._host_wait(1, slot, self._seq[slot])             # the consumer is done with the batch this slot held (normally long since)
        seq = self._seq[slot] = self._staged_count = getattr(self, "_staged_count", 0) + 1
        with torch.cuda.stream(self._stream):
            _mark("upload_begin")
            out = {k: (v if (self.device_aug is not None and k in self._RAW) else self._move(bufs, k, v)) for k, v in batch.items()}
            if self.device_aug is not None:
                out = self.device_aug(out, bufs)
            _mark("upload_end")
            self._signal(0, slot, seq)
        return out, seq

    def _pick_stream(self, n):
        """Copy-stream selection (see __init__): n = batches handed over so far, across re-iterations."""
        import time
        p = self._pick
        p["count"] = k = p.get("count", 0) + 1
        now = time.perf_counter()
        if k in (5, 21):
            p["t"] = [now]
        elif 5 < k <= 16 or 21 < k <= 32:
            p["t"].append(now)
        if k == 16 or k == 32:
            d = sorted(b - a for a, b in zip(p["t"], p["t"][1:]))
            p["med"][p["phase"]] = d[len(d) // 2]
            if k == 16:
                p["first"], p["phase"] = self._stream, 1
                self._stream = torch.cuda.Stream(device=self.device)
            else:
                if p["med"][1] > 0.99 * p["med"][0]:           # the fresh stream is not clearly better: back to the first one
                    self._stream = p["first"]
                self.picked = {"first_ms": p["med"][0] * 1e3, "fresh_ms": p["med"][1] * 1e3, "kept": "first" if self._stream is p["first"] else "fresh"}
                self._pick = None

    def _signal(self, which, slot, seq):
        """Enqueue on the current stream: flags[which][slot] = seq once everything enqueued before has completed."""
        from ..hip import lib as L
        L.check(L.lib().sde_store_u64(ctypes.c_void_p(self._flags.data_ptr() + 8 * (which * self._slots + slot)), seq, L.stream()), "sde_store_u64")

    def _host_wait(self, which, slot, seq):
        """Spin until flags[which][slot] >= seq (pinned memory the device writes; no HIP call, nothing enqueued on any stream)."""
        if seq <= 0:
            return
        f = self._flags[which]
        while int(f[slot]) < seq:
            pass

    def __iter__(self):
        from collections import deque
        it = iter(self.loader)
        staged, n_staged, n = deque(), 0, 0
        def stage_more():
            nonlocal n_staged, it
            while it is not None and len(staged) < self._ahead:          # (+ the batch in use = at most `ahead + 1` <= slots buffer sets alive)
                try:
                    b = next(it)
                except StopIteration:
                    it = None
                    return
                staged.append(self._stage(b, n_staged % self._slots))
                n_staged += 1
        stage_more()
        while staged:
            cur, seq = staged.popleft()
            slot = n % self._slots
            n += 1
            if self._pick is not None:
                self._pick_stream(n)
            stage_more()                            # the uploads of the next `ahead` batches are in flight before this one is handed over
            if seq is not None:
                self._host_wait(0, slot, seq)       # the upload has completed (the host saw its flag): the consumer's stream needs no wait of its own
            yield cur
            if seq is not None:                     # the consumer came back: everything it enqueued on this slot's tensors is ahead of this flag store
                self._signal(1, slot, seq)
