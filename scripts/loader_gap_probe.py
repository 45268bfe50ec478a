"""Which part of the --with-loader step costs the time the kernels do not account for: variants of the prefetcher's staging (diagnostic, wrong data
on purpose in two of them): normal | nocopy (slot buffers reused, no uploads, no events) | noevent (uploads issued, consumer does not wait for them)."""
import sys
import time

import torch

sys.path.insert(0, ".")
import bench  # noqa: E402


def run(workload, variant, n_warm=20, n=120):
    class A:
        dtype, no_graph, force_overlap, no_pose_stream = "bf16", False, False, False
    A.workload = workload
    dev = torch.device("cuda", 0)
    cfg, model, trainer = bench.build(A, dev)
    from simpledepthestimation_amd.data import DevicePrefetcher
    from simpledepthestimation_amd.data.device_aug import DeviceImageAug
    arch = bench.WORKLOADS[workload]["arch"]
    trainer.input_transform = DeviceImageAug(dev)
    gen, _ = bench.host_loader(arch, 12, 192, 640, 1, 2 * n_warm + n)
    import os
    skipped = [torch.cuda.Stream() for _ in range(int(os.environ.get("SKIP_STREAMS", "0")))]      # shift the pool index the copy stream gets
    pf = DevicePrefetcher(gen, dev, slots=int(os.environ.get("SLOTS", "4")), ahead=int(os.environ.get("AHEAD", "2")))
    if variant == "priostream":          # copy stream from the high-priority pool from the start (before the graph exists)
        pf._stream = torch.cuda.Stream(device=dev, priority=-1)
    orig = pf._stage
    cache = {}
    def stage(batch, slot):
        if variant not in ("nocopy", "noevent") or slot not in cache:
            out, ev = orig(batch, slot)
            cache[slot] = out
            return out, (ev if variant != "noevent" else None)
        if variant == "nocopy":
            return cache[slot], None
        out, ev = orig(batch, slot)
        return out, None
    pf._stage = stage
    if variant.startswith("chunk"):
        # the full upload, in `chunks` pieces per entry
        nchunk = int(variant[5:])
        real_into2 = pf._into
        def into2(bufs, key, v):
            b = bufs.get(key)
            if b is not None and torch.is_tensor(v) and v.dim() == 4 and b.shape == v.shape:
                per = -(-v.shape[0] // nchunk)
                for i in range(0, v.shape[0], per):
                    b[i:i + per].copy_(v[i:i + per], non_blocking=True)
                return b
            return real_into2(bufs, key, v)
        pf._into = into2
    if variant in ("smallcopy", "halfcopy"):
        # every upload still issued, but of one frame (of six) per entry only: does the cost follow the BYTES or the NUMBER of copies?
        real_into = pf._into
        def into(bufs, key, v):
            b = bufs.get(key)
            if b is not None and torch.is_tensor(v) and v.dim() == 4 and b.shape == v.shape:
                k = 1 if variant == "smallcopy" else v.shape[0] // 2
                b[:k].copy_(v[:k], non_blocking=True)
                return b
            return real_into(bufs, key, v)
        pf._into = into
    if variant in ("hostsync", "query"):
        # the consumer's cross-stream wait replaced by a host-side wait on (or query of) the upload's event: nothing is enqueued on the main stream
        class Ev:
            def __init__(self, e): self.e = e
        real_wait = torch.cuda.Stream.wait_event
        def wait_event(self_, e):
            if self_ == torch.cuda.current_stream() and getattr(e, "_upload", False):
                if variant == "hostsync":
                    e.synchronize(); return
                if e.query():
                    return
            return real_wait(self_, e)
        torch.cuda.Stream.wait_event = wait_event
        def stage2(batch, slot):
            out, ev = orig(batch, slot)
            if ev is not None:
                ev._upload = True
            return out, ev
        pf._stage = stage2
    if variant in ("relquery", "relquery+query"):
        # release events still recorded on the main stream, but the copy stream waits on one only if it has not completed yet (host-side query)
        real_wait2 = torch.cuda.Stream.wait_event
        rels = set()
        class Rel(list):
            def __setitem__(self, i, v):
                rels.add(id(v)); v._rel = True
                list.__setitem__(self, i, v)
        pf._release = Rel(pf._release)
        def wait_event2(self_, e):
            if getattr(e, "_rel", False) and e.query():
                return
            if variant == "relquery+query" and getattr(e, "_upload", False) and e.query():
                return
            return real_wait2(self_, e)
        torch.cuda.Stream.wait_event = wait_event2
        def stage3(batch, slot):
            out, ev = orig(batch, slot)
            if ev is not None:
                ev._upload = True
            return out, ev
        pf._stage = stage3
    if variant == "norelease+query":
        real_wait3 = torch.cuda.Stream.wait_event
        cnt = {"skipped": 0, "waited": 0}
        def wait_event3(self_, e):
            if getattr(e, "_upload", False) and e.query():
                cnt["skipped"] += 1
                return
            cnt["waited"] += 1
            return real_wait3(self_, e)
        torch.cuda.Stream.wait_event = wait_event3
        def stage4(batch, slot):
            out, ev = orig(batch, slot)
            if ev is not None:
                ev._upload = True
            return out, ev
        pf._stage = stage4
        import atexit
        atexit.register(lambda: print("   upload waits skipped / enqueued:", cnt))
    if variant in ("norelease", "norelease+query"):
        class NoRel(list):
            def __setitem__(self, i, v): pass
        pf._release = NoRel(pf._release)
    feed = iter(pf)
    for _ in range(n_warm):
        trainer.step(next(feed))
    torch.cuda.synchronize()
    if variant in ("restream", "rebuf", "repin"):
        # finer: only the copy stream / only the device slot buffers / only the pinned host batch replaced after the graph exists
        if variant == "restream":
            pf._stream = torch.cuda.Stream(device=dev)
        elif variant == "rebuf":
            pf._bufs = [dict() for _ in range(pf._slots)]
        else:
            gen2, _ = bench.host_loader(arch, 12, 192, 640, 3, 1)
            fresh = next(iter(gen2))
            pf.loader = (dict(fresh) for _ in range(n_warm + n))
            it2 = iter(pf)
            feed = it2
        for _ in range(n_warm):
            trainer.step(next(feed))
        torch.cuda.synchronize()
    if variant in ("reprefetch", "reprefetch_keep", "recapture"):
        # which object of the first run carries the slow state?  reprefetch: same trainer, NEW host buffers + prefetcher (old ones released first);
        # reprefetch_keep: new ones while the old ones stay alive; recapture: same prefetcher, the trainer's graph captured again
        if variant == "recapture":
            trainer._graphs = {}; trainer._static_batch = None
        else:
            if variant == "reprefetch":
                del feed, pf, gen
                import gc; gc.collect()
            else:
                keep = (feed, pf, gen)
            gen, _ = bench.host_loader(arch, 12, 192, 640, 2, n_warm + n)
            pf = DevicePrefetcher(gen, dev)
            feed = iter(pf)
        for _ in range(n_warm):
            trainer.step(next(feed))
        torch.cuda.synchronize()
    t0 = time.perf_counter()
    chunks, tc, i = [], t0, 0
    for hb in feed:
        trainer.step(hb)
        i += 1
    torch.cuda.synchronize()
    print("   per 40 steps:", chunks, flush=True)
    from simpledepthestimation_amd.hip import lib as L
    if L.MARKS is not None:
        print("   marks:", {k: round(v) for k, v in sorted(L.marks_read("step_start").items(), key=lambda kv: kv[1])}, flush=True)
    return (time.perf_counter() - t0) / max(i, 1) * 1e3


wl = sys.argv[1] if len(sys.argv) > 1 else "mono_r18"
import os
if os.environ.get("MARKS"):
    from simpledepthestimation_amd.hip import lib as L
    L.marks_enable(torch.device("cuda", 0))
for v in sys.argv[2:] or ["normal", "nocopy", "noevent"]:
    print(f"{wl} {v}: {run(wl, v):.3f} ms/step", flush=True)
