"""Is host time between two graph replays exposed on this stack?  Resident-input steps with a busy-wait of D microseconds per iteration."""
import sys
import time

import torch

sys.path.insert(0, ".")
import bench  # noqa: E402


class A:
    workload, dtype, no_graph, force_overlap, no_pose_stream = "sup_r50", "bf16", False, False, False


dev = torch.device("cuda", 0)
cfg, model, trainer = bench.build(A, dev)
batch = bench.synth_batch("SupDepthModel", 12, 192, 640, 5, dev)
for _ in range(10):
    trainer.step(batch)
for delay_us in (0, 200, 400, 800, 0):
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(60):
        t = time.perf_counter()
        while (time.perf_counter() - t) * 1e6 < delay_us:
            pass
        trainer.step(batch)
    torch.cuda.synchronize()
    print(f"busy-wait {delay_us:4d} us per iteration: {(time.perf_counter() - t0) / 60 * 1e3:.3f} ms/step")
# three replays back to back after a sync: how long does each launch call take on the host?
torch.cuda.synchronize()
for i in range(4):
    t = time.perf_counter()
    trainer._graph.replay()
    print(f"replay {i}: host {(time.perf_counter() - t) * 1e3:.3f} ms")
torch.cuda.synchronize()
