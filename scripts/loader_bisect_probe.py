"""Bisect the --with-loader cost: which part of the input side slows the replayed step (scripts run on the GPU box)."""
import sys
import time

import torch

sys.path.insert(0, ".")
import bench  # noqa: E402
from simpledepthestimation_amd.data.device_aug import DeviceImageAug  # noqa: E402


class A:
    workload, dtype, no_graph, force_overlap, no_pose_stream = "sup_r50", "bf16", False, False, False


dev = torch.device("cuda", 0)
cfg, model, trainer = bench.build(A, dev)
aug = DeviceImageAug(dev)
gen, _ = bench.host_loader("SupDepthModel", 12, 192, 640, 1, 1)
hb = next(iter(gen))
frames_h, params_h, depth_h = hb["img_u8"], hb["aug_params"], hb["depth"]
frames_d, params_d, depth_d = frames_h.to(dev), params_h.to(dev), depth_h.to(dev)
side = torch.cuda.Stream()
bufs = [dict() for _ in range(3)]
img, orig = aug.prep(frames_d, params_d, 192, 640)
batch = {"img": img.clone(), "img_orig": orig.clone(), "depth": depth_d.clone()}
for _ in range(10):
    trainer.step(batch)


def run(name, work, on_side=True, n=60):
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for i in range(n):
        if work is not None:
            if on_side:
                with torch.cuda.stream(side):
                    work(bufs[i % 3])
            else:
                work(bufs[i % 3])
        trainer.step(batch)
    torch.cuda.synchronize()
    print(f"{name:58s} {(time.perf_counter() - t0) / n * 1e3:.3f} ms/step")


def h2d(b):
    aug._up(frames_h, b, "img"); aug._up(depth_h, b, "depth")


def prep(b):
    aug.prep(frames_d, params_d, 192, 640, b, "img")


def both(b):
    aug.prep(aug._up(frames_h, b, "img"), params_d, 192, 640, b, "img"); aug._up(depth_h, b, "depth")


def d2d(b):
    aug._buf(b, "x", frames_d.shape, frames_d.dtype, dev).copy_(frames_d)


def run_ev(name, main_waits, side_waits, lag, n=60):
    """the prefetcher's event pattern: work for step i+lag is issued before step i; main waits for its batch's event; side waits for a release event"""
    torch.cuda.synchronize()
    evs, rels = {}, {}
    t0 = time.perf_counter()
    for i in range(n + lag):
        j = i                      # stage batch j
        if j < n:
            with torch.cuda.stream(side):
                if side_waits and (j - 3) in rels:
                    side.wait_event(rels[j - 3])
                both(bufs[j % 3])
                e = torch.cuda.Event(); e.record(side); evs[j] = e
        k = i - lag                # run step k
        if k >= 0:
            if main_waits:
                torch.cuda.current_stream().wait_event(evs[k])
            trainer.step(batch)
            r = torch.cuda.Event(); r.record(torch.cuda.current_stream()); rels[k] = r
    torch.cuda.synchronize()
    print(f"{name:58s} {(time.perf_counter() - t0) / n * 1e3:.3f} ms/step")


run("resident, nothing beside the step", None)
run_ev("side work, lag 1, no events", False, False, 1)
run_ev("side work, lag 1, main waits for the batch event", True, False, 1)
run_ev("side work, lag 1, side waits for the release event", False, True, 1)
run_ev("side work, lag 1, both waits (the prefetcher)", True, True, 1)
run_ev("side work, lag 2, both waits", True, True, 2)
run("H2D of 22 MB on a side stream", h2d)
run("device-to-device copy of 17 MB on a side stream", d2d)
run("prep kernels (device-resident frames) on a side stream", prep)
run("H2D + prep on a side stream", both)
run("H2D + prep on the MAIN stream (nothing overlaps)", both, on_side=False)
run("resident, nothing beside the step", None)


from simpledepthestimation_amd.data import DevicePrefetcher  # noqa: E402


def run_pf(name, mode, slots=3, n=60):
    """mode: 'fixed' = the prefetcher runs beside steps on a fixed resident batch; 'copy' = aug on the copy stream, step consumes the batch (D2D copies into the
    static inputs); 'inplace' = uploads only, HipTrainer.input_transform writes the static inputs"""
    gen, _ = bench.host_loader("SupDepthModel", 12, 192, 640, 1, n)
    trainer.input_transform = aug if mode == "inplace" else None
    pf = DevicePrefetcher(gen, dev, device_aug=(None if mode == "inplace" else aug), slots=slots)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for hb_ in pf:
        trainer.step(batch if mode == "fixed" else hb_)
    torch.cuda.synchronize()
    trainer.input_transform = None
    print(f"{name:58s} {(time.perf_counter() - t0) / n * 1e3:.3f} ms/step")


for rep in range(3):
    run_pf("prefetcher beside steps on a FIXED batch", "fixed")
    run_pf("prefetcher, aug on copy stream, step consumes batch", "copy")
    run_pf("prefetcher uploads, transform in place (bench form)", "inplace")
    run_pf("  ... the same with 4 slots", "inplace", slots=4)
    run("resident, nothing beside the step", None)
