"""Host time per phase of the --with-loader step in bench.py's configuration (resize / jitter kernels at the head of the step, writing the graph's static
inputs): next(feed) = the prefetcher's staging, input_transform, graph replay, optimizer launch.  Wall clock around each call, no device sync inside."""
import sys
import time

import torch

sys.path.insert(0, ".")
import bench  # noqa: E402


def main():
    class A:
        workload, dtype, no_graph, force_overlap, no_pose_stream = sys.argv[1] if len(sys.argv) > 1 else "mono_r18", "bf16", False, False, False
    dev = torch.device("cuda", 0)
    cfg, model, trainer = bench.build(A, dev)
    from simpledepthestimation_amd.data import DevicePrefetcher
    from simpledepthestimation_amd.data.device_aug import DeviceImageAug
    arch = bench.WORKLOADS[A.workload]["arch"]
    aug = DeviceImageAug(dev)
    trainer.input_transform = aug
    n_warm, n = 20, 60
    gen, _ = bench.host_loader(arch, 12, 192, 640, 1, n_warm + n)
    feed = iter(DevicePrefetcher(gen, dev))
    for _ in range(n_warm):
        trainer.step(next(feed))
    torch.cuda.synchronize()
    acc = {}
    def wrap(obj, name, label):
        f = getattr(obj, name)
        def g(*a, **k):
            t = time.perf_counter()
            r = f(*a, **k)
            acc[label] = acc.get(label, 0.0) + time.perf_counter() - t
            return r
        setattr(obj, name, g)
    wrap(trainer, "input_transform", "input_transform")
    wrap(trainer, "_copy_into_static", "copy_into_static")
    wrap(trainer, "_optimizer", "optimizer")
    for k_, (ga, gb, so) in list(trainer._graphs.items()):
        class GW:
            def __init__(self, g): self.g = g
            def replay(self):
                t = time.perf_counter(); self.g.replay(); acc["graph.replay"] = acc.get("graph.replay", 0.0) + time.perf_counter() - t
        trainer._graphs[k_] = (GW(ga), gb, so)
    t0 = time.perf_counter()
    tn = 0.0
    for _ in range(n):
        t = time.perf_counter(); hb = next(feed); tn += time.perf_counter() - t
        trainer.step(hb)
    t1 = time.perf_counter()
    torch.cuda.synchronize()
    t2 = time.perf_counter()
    print(f"{A.workload}: host {(t1 - t0) / n * 1e3:.3f} ms/step, with final sync {(t2 - t0) / n * 1e3:.3f}; next(feed) {tn / n * 1e3:.3f}")
    print({k: round(v / n * 1e3, 3) for k, v in acc.items()})
    # the same trainer fed from device-resident inputs (what bench.py times without --with-loader)
    acc.clear()
    trainer.input_transform = None
    batch = {k: ([x.clone() for x in v] if isinstance(v, list) else (v.clone() if torch.is_tensor(v) else v)) for k, v in trainer._static_batch.items()}
    for _ in range(5):
        trainer.step(batch)
    torch.cuda.synchronize()
    acc.clear()
    t0 = time.perf_counter()
    for _ in range(n):
        trainer.step(batch)
    t1 = time.perf_counter()
    torch.cuda.synchronize()
    t2 = time.perf_counter()
    print(f"{A.workload} resident: host {(t1 - t0) / n * 1e3:.3f} ms/step, with final sync {(t2 - t0) / n * 1e3:.3f}")
    print({k: round(v / n * 1e3, 3) for k, v in acc.items()})


main()
