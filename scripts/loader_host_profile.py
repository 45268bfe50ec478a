"""Where the --with-loader step spends host time: staging only, staging + step, and a cProfile of the loop (scripts/gpu_r03_loaderprof.sh)."""
import cProfile
import pstats
import sys
import time

import torch

sys.path.insert(0, ".")
import bench  # noqa: E402


def main():
    class A:
        workload, dtype, no_graph, force_overlap, no_pose_stream = sys.argv[1] if len(sys.argv) > 1 else "sup_r50", "bf16", False, False, False
    dev = torch.device("cuda", 0)
    cfg, model, trainer = bench.build(A, dev)
    from simpledepthestimation_amd.data import DevicePrefetcher
    from simpledepthestimation_amd.data.device_aug import DeviceImageAug
    arch = bench.WORKLOADS[A.workload]["arch"]
    aug = DeviceImageAug(dev)

    def loop(n, step=True):
        gen, _ = bench.host_loader(arch, 12, 192, 640, 1, n)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for hb in DevicePrefetcher(gen, dev, device_aug=aug):
            if step:
                trainer.step(hb)
        t1 = time.perf_counter()
        torch.cuda.synchronize()
        t2 = time.perf_counter()
        return (t1 - t0) / n * 1e3, (t2 - t0) / n * 1e3
    loop(10)
    # host time per call site (wall clock around the call: an entry that is large means the HOST blocks there)
    acc = {}
    def wrap(obj, name, label):
        f = getattr(obj, name)
        def g(*a, **k):
            t = time.perf_counter()
            r = f(*a, **k)
            acc[label] = acc.get(label, 0.0) + time.perf_counter() - t
            return r
        setattr(obj, name, g)
    wrap(DevicePrefetcher, "_stage", "prefetcher._stage")
    wrap(trainer, "_copy_into_static", "trainer._copy_into_static")
    wrap(trainer, "_optimizer", "trainer._optimizer")
    g0 = trainer._graph
    class GW:
        def replay(self):
            t = time.perf_counter(); g0.replay(); acc["graph.replay"] = acc.get("graph.replay", 0.0) + time.perf_counter() - t
    key = trainer._graph_key({"flip": False}) if False else None
    for k_, (ga, gb, so) in list(trainer._graphs.items()):
        trainer._graphs[k_] = (GW(), gb, so)
    print("with loader   : host %.3f ms/step, with sync %.3f" % loop(40))
    print({k: round(v / 40 * 1e3, 3) for k, v in acc.items()})
    acc.clear()
    batch = {k: (v.clone() if torch.is_tensor(v) else v) for k, v in trainer._static_batch.items()}
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(40):
        trainer.step(batch)
    t1 = time.perf_counter(); torch.cuda.synchronize()
    print("resident      : host %.3f ms/step, with sync %.3f" % ((t1 - t0) / 40 * 1e3, (time.perf_counter() - t0) / 40 * 1e3))
    print({k: round(v / 40 * 1e3, 3) for k, v in acc.items()})
    print("staging only  : host %.3f ms/step, with sync %.3f" % loop(40, step=False))
    print("staging + step: host %.3f ms/step, with sync %.3f" % loop(40))
    if len(sys.argv) > 2 and sys.argv[2] == "short":
        return
    pr = cProfile.Profile()
    pr.enable()
    loop(40)
    pr.disable()
    pstats.Stats(pr).sort_stats("cumulative").print_stats(25)


main()
