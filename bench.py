#!/usr/bin/env python
"""bench.py -- headline metric of BASELINE.json: training images/sec at 192x640, bs=12/GPU.

    python bench.py --gpus 1 --steps 20 --warmup 5
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 --master-port P bench.py --gpus N ...
    python bench.py --gpus N ...        (no launcher: the parent starts the N ranks itself, as detectron2/engine/launch.py:L24-92 does)

Workload (N=1 and per rank for N>1): BASELINE.json configs[1] -- Supervised ResNet-50, bf16 storage / fp32 accumulate,
bs=12, 192x640, synthetic KITTI-shaped batches, random-init weights.  One "step" = zero-grad + forward +
backward (+ RCCL all-reduce of the flat gradient for N>1) + fused AdamW, exactly what projects/Supervised/train.py:L99-128 does.
Inputs are resident in HBM when the timed region starts.  Rank 0 prints ONE JSON line.

Extra objects on that line (tier contract):
  roofline     -- dominant kernel family by device time (the implicit-GEMM convolution kernel), algorithmic FLOPs / measured
                  duration, from an event-instrumented pass of the same step inside this process (events on the launch stream).
  cpu_baseline -- the CPU oracle (torch-CPU restatement of the reference, "port") timed on this host on a bounded sample.
"""
import argparse
import json
import os
import subprocess
import sys
import time

import torch
import torch.distributed as dist

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

HBM_PEAK_GBPS = 8000.0
MFMA_PEAK_TFLOPS = {"bf16": 2500.0, "fp16": 2500.0, "fp32": 157.3}    # dense peaks, /opt/skills/guides/MI355X_MICROARCH.md
WORKLOADS = {
    "sup_r50": dict(arch="SupDepthModel", enc="50", desc="Supervised ResNet-50 (BASELINE configs[1])"),
    "sup_r18": dict(arch="SupDepthModel", enc="18", desc="Supervised ResNet-18"),
    "mono_r18": dict(arch="MonoDepth2Model", enc="18", desc="MonoDepth2 ResNet-18 3-frame (BASELINE configs[2])"),
    "mono_r50": dict(arch="MonoDepth2Model", enc="50", desc="MonoDepth2 ResNet-50 3-frame (BASELINE configs[3])"),
    "mono_packnet": dict(arch="MonoDepth2Model", enc="18", packnet="1A", desc="MonoDepth2 PackNet-1A 3-frame (BASELINE configs[4]; --dtype fp16 = its fp16 + loss scaling)"),
}


def synth_batch(arch, B, H, W, seed, device):
    from simpledepthestimation_amd.data.synthetic import mono_batch, sup_batch
    batch = sup_batch(B, H, W, seed) if arch == "SupDepthModel" else mono_batch(B, H, W, seed)
    return {k: ([x.to(device) for x in v] if isinstance(v, list) else v.to(device)) for k, v in batch.items()}


def build(args, device):
    from simpledepthestimation_amd.config import get_project_cfg
    from simpledepthestimation_amd.engine import trainer as T
    from simpledepthestimation_amd.modeling import build_model
    wl = WORKLOADS[args.workload]
    cfg = get_project_cfg("Supervised" if wl["arch"] == "SupDepthModel" else "MonoDepth2")
    cfg.MODEL.META_ARCHITECTURE = wl["arch"]
    cfg.MODEL.DEPTH_NET.ENCODER_NAME = wl["enc"]
    cfg.MODEL.COMPUTE_DTYPE = args.dtype
    cfg.SOLVER.AMP = args.dtype == "fp16"
    cfg.MODEL.DEVICE = str(device)
    if wl.get("packnet"):                           # projects/MonoDepth2/configs/packnet_1a.yaml
        cfg.MODEL.DEPTH_NET.NAME, cfg.MODEL.DEPTH_NET.VERSION, cfg.LOSS.VAR_LOSS_WEIGHT = "PackNet01", wl["packnet"], 1e-4
    torch.manual_seed(0)
    model = build_model(cfg).train()
    mk = T.supervised_trainer if wl["arch"] == "SupDepthModel" else T.monodepth2_trainer
    trainer = mk(model, cfg, use_graph=not args.no_graph, overlap=(True if args.force_overlap else None), pose_stream=not args.no_pose_stream)
    return cfg, model, trainer


def host_loader(arch, B, H, W, seed, steps):
    """--with-loader: what a KITTI data loader with the ON_DEVICE preprocess chain hands over per step -- uint8 camera frames at source size
    (375 x 1242), stacked per entry and pinned (the collator's and the pin-memory thread's work), the drawn colour-jitter parameters, and the small
    fp32 entries (intrinsics / sparse depth) -- yielded `steps` times.  Decoding PNGs is the workers' job and not part of this measurement; the
    host-to-device copies, the resize + colour-jitter kernels and the copy into the graph's static inputs are."""
    import numpy as np
    from simpledepthestimation_amd.data.synthetic import mono_batch, sup_batch
    rng = np.random.default_rng(seed)
    Hs, Ws = 375, 1242
    small = sup_batch(B, H, W, seed) if arch == "SupDepthModel" else mono_batch(B, H, W, seed)
    frames = lambda: torch.from_numpy(rng.integers(0, 256, (B, Hs, Ws, 3), dtype=np.uint8)).pin_memory()
    params = []
    for _ in range(B):
        params.append([float(rng.uniform(0.8, 1.2)), float(rng.uniform(0.8, 1.2)), float(rng.uniform(0.8, 1.2)), float(rng.uniform(-0.05, 0.05))] + [float(i) for i in rng.permutation(4)])
    batch = {"img_u8": frames(), "aug_params": torch.tensor(params, dtype=torch.float32).pin_memory(), "device_resize": (H, W)}
    if arch == "SupDepthModel":
        batch["depth"] = small["depth"].pin_memory()
    else:
        batch["ctx_img_u8"] = [frames(), frames()]
        batch["intrinsics"] = small["intrinsics"].pin_memory()
    nbytes = sum(t.numel() * t.element_size() for v in batch.values() for t in (v if isinstance(v, list) else [v]) if torch.is_tensor(t))
    return (dict(batch) for _ in range(steps)), nbytes


def roofline_pass(trainer, batch, steps, dtype, workload="sup_r50", step_ms=None):
    """Event-instrumented eager steps of the same workload: per GEMM launch (kind, tile variant) duration and algorithmic FLOPs.

    Every eager step issues the same launches in the same order, so a launch is identified by its index inside the step and its duration is
    the MINIMUM over the `steps` instrumented steps (>= 4): an event pair also times whatever the host did between its launches, and one
    host hiccup inside one pair used to poison a two-step sum (BENCH_r02: 7.68 ms "inside" a 6.73 ms step).  The line is flagged
    `roofline_suspect` when it still contradicts the timed region."""
    from simpledepthestimation_amd.hip import lib as L
    steps = max(4, steps)
    def one():
        trainer._fwd_bwd(batch)
        if trainer._cut is not None:
            trainer._backward_rest()
    one()                                        # eager warm-up (the timed region may have run under graph replay)
    torch.cuda.synchronize()
    per_step = []
    for _ in range(steps):
        L.PROFILE, L.PROFILE_REPEAT = [], 6      # each GEMM launch six times back to back between its two events (lib.timed)
        one()
        torch.cuda.synchronize()
        per_step.append([(kind, work, variant, e0.elapsed_time(e1) / rep, meta) for kind, work, variant, e0, e1, meta, rep in L.PROFILE])
    L.PROFILE, L.PROFILE_REPEAT = None, 1
    n = len(per_step[0])
    same = all(len(s) == n and all(a[:3] == b[:3] for a, b in zip(s, per_step[0])) for s in per_step)
    if same:                                     # (kind, work, variant, min ms, median ms, meta) per launch of ONE step
        recs = []
        for i in range(n):
            ds = sorted(s[i][3] for s in per_step)
            recs.append((per_step[0][i][0], per_step[0][i][1], per_step[0][i][2], ds[0], ds[len(ds) // 2], per_step[0][i][4]))
    else:                                        # launch sequences differ between steps (not expected): fall back to the last step as it is
        recs = [(k, w, v, ms, ms, m) for k, w, v, ms, m in per_step[-1]]
    if os.environ.get("SDE_BENCH_LAYER_DUMP"):
        with open(os.environ["SDE_BENCH_LAYER_DUMP"], "w") as f:
            f.write("kind,variant,M,N,K,k,stride,mode,splits,us,tflops,unique_GBps\n")
            for kind, flops, variant, ms, _med, meta in recs:
                if kind.startswith("photo"):
                    continue
                us = ms * 1e3
                m = meta or {}
                f.write(f"{kind},{variant},{m.get('M')},{m.get('N')},{m.get('K')},{m.get('k')},{m.get('s')},{m.get('mode')},{m.get('splits', '')},"
                        f"{us:.1f},{flops / us / 1e6:.1f},{m.get('bytes', 0) / us / 1e3:.0f}\n")
    photo = {}
    for kind, nbytes, variant, ms, _med, meta in recs:
        if kind.startswith("photo"):
            f = photo.setdefault((kind, meta["h"], meta["w"]), {"ms": 0.0, "bytes": 0.0, "launches": 0})
            f["ms"] += ms; f["bytes"] += nbytes; f["launches"] += 1
    all_recs = recs
    recs = [r for r in recs if not r[0].startswith("photo")]
    hbm = None
    if photo:
        k0 = max((k for k in photo if k[0] == "photo_fwd"), key=lambda k: k[1] * k[2])        # the full-resolution scale
        v = photo[k0]
        gbps = v["bytes"] / (v["ms"] * 1e-3) / 1e9
        nsc = next((m.get("scales") for k_, _w, _v, _ms, _med, m in [(r[0], r[1], r[2], r[3], r[4], r[5]) for r in all_recs] if k_ == "photo_fwd" and m), None)
        where = f"all {nsc} scales from {k0[1]}x{k0[2]} down in one launch" if nsc else f"scale {k0[1]}x{k0[2]}"
        hbm = {"bound": "hbm", "kernel": f"photo_fwd_kernel (warp + SSIM + L1 + min/automask), {where}", "achieved": round(gbps, 1),
               "peak": HBM_PEAK_GBPS, "unit": "GB/s", "frac": round(gbps / HBM_PEAK_GBPS, 4), "traffic": None,
               "avg_launch_us": round(v["ms"] * 1e3 / v["launches"], 2),
               "all": {f"{k[0]}:{k[1]}x{k[2]}": {"us": round(x["ms"] * 1e3 / x["launches"], 2), "GBps": round(x["bytes"] / (x["ms"] * 1e-3) / 1e9, 1)}
                       for k, x in sorted(photo.items(), key=lambda kv: -kv[0][1])}}
    fam = {}
    peak = MFMA_PEAK_TFLOPS[dtype]
    roof_ms = 0.0                                # sum over launches of the roofline time: max(FLOPs / MFMA peak, algorithmic bytes / HBM peak)
    for kind, flops, variant, ms, med, meta in recs:
        key = ("igemm" if kind.startswith("igemm") else kind, variant)
        f = fam.setdefault(key, {"ms": 0.0, "med": 0.0, "flops": 0.0, "launches": 0, "bytes": 0.0})
        f["ms"] += ms; f["med"] += med; f["flops"] += flops; f["launches"] += 1; f["bytes"] += (meta or {}).get("bytes", 0)
        if flops > 0:
            roof_ms += max(flops / (peak * 1e12), (meta or {}).get("bytes", 0) / (HBM_PEAK_GBPS * 1e9)) * 1e3
    gemm = {k: v for k, v in fam.items() if v["flops"] > 0}
    dom_key = max(gemm, key=lambda k: gemm[k]["ms"])
    dom = gemm[dom_key]
    achieved = dom["flops"] / (dom["ms"] * 1e-3) / 1e12
    tot_ms = sum(f["ms"] for f in gemm.values()); tot_fl = sum(f["flops"] for f in gemm.values())
    if dom_key[0] != "igemm":
        name = "wgrad_dma_kernel / wgrad_kernel<16-bit,64x128>" if dtype != "fp32" else "wgrad_kernel<f32>"
    elif dom_key[1] >= 7000000:
        name = f"pgemm_kernel<{dtype},{(dom_key[1] - 7000000) // 1000}x{dom_key[1] % 1000}> (persistent LDS-DMA GEMM)"
    elif dom_key[1] >= 3000000:
        name = f"halo3_kernel<{dtype},8x16 pixels x {dom_key[1] % 1000}>"
    else:
        name = f"igemm_kernel<{dtype},{dom_key[1] // 1000}x{dom_key[1] % 1000}>"
    traffic, traffic_src = pmc_traffic(dom_key, dtype) if workload == "sup_r50" else (None, None)      # the committed counter passes are of that workload
    out = {"bound": "mfma", "achieved": round(achieved, 2), "peak": peak, "unit": "TFLOP/s", "frac": round(achieved / peak, 4), "traffic": traffic,
           "traffic_source": traffic_src,
           "kernel": name, "launches_per_step": dom["launches"], "avg_launch_us": round(dom["ms"] * 1e3 / dom["launches"], 2),
           "median_launch_us": round(dom["med"] * 1e3 / dom["launches"], 2),
           "algorithmic_GBps": round(dom["bytes"] / (dom["ms"] * 1e-3) / 1e9, 1),
           "timing": f"per launch: min over {steps} instrumented eager steps of (event pair around 6 back-to-back launches) / 6",
           "gemm_flops_per_step": tot_fl, "gemm_ms_per_step": round(tot_ms, 3),
           "all_gemm_achieved": round(tot_fl / (tot_ms * 1e-3) / 1e12, 2),
           # fraction of the per-launch roofline (each launch against the roof that binds IT: MFMA peak or HBM peak on its algorithmic bytes)
           "all_gemm_attainable_frac": round(roof_ms / tot_ms, 4),
           "families": {f"{k[0]}:{k[1]}": {"ms_per_step": round(v["ms"], 3), "tflops": round(v["flops"] / (v["ms"] * 1e-3) / 1e12, 2),
                                           "launches_per_step": v["launches"]} for k, v in sorted(gemm.items(), key=lambda kv: -kv[1]["ms"])}}
    # sanity guards: a GEMM family cannot take longer than the whole timed step, and the median must agree with the minimum
    suspect = []
    if step_ms is not None and any(v["ms"] > step_ms for v in gemm.values()):
        suspect.append("a family's ms_per_step exceeds ms_per_step of the timed region")
    if step_ms is not None and tot_ms > 2.0 * step_ms:
        suspect.append("gemm_ms_per_step exceeds twice the timed step (the serial kernel sum is ~1.4x the two-queue step)")
    if dom["med"] > 1.25 * dom["ms"]:
        suspect.append("median launch duration of the dominant family is > 1.25x its minimum")
    if not same:
        suspect.append("launch sequences differed between instrumented steps")
    out["roofline_suspect"] = bool(suspect)
    if suspect:
        out["roofline_suspect_why"] = suspect
    return hbm, out


def pmc_traffic(dom_key, dtype):
    """HBM bytes per launch of the dominant kernel family from the committed counter passes of the same command (scripts/gpu_profile.sh:
    separate rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE runs, kernel-trace only; FETCH_SIZE doubled for 16-byte-per-lane reads as
    MI355X_MICROARCH.md prescribes; KB -> bytes).  None when no matching profile is committed (e.g. other dtype / workload)."""
    import csv
    import glob
    if dtype == "fp32":
        return None, None
    kind, variant = dom_key
    if kind == "igemm":
        if variant >= 7000000:
            pat = f"pgemm_kernelIDF16{'b' if dtype == 'bf16' else '_'}Li{(variant - 7000000) // 1000}ELi{variant % 1000}E"
        else:
            pat = "halo3_kernel" if variant >= 3000000 else f"igemm_kernelIDF16bLi{variant // 1000}ELi{variant % 1000}E"
    elif kind == "wgrad":
        pat = "wgrad_kernelIDF16b"
    else:
        return None, None
    fetch = sorted(glob.glob(os.path.join(ROOT, "profiles", "r*_fetch_by_kernel.csv")))
    write = sorted(glob.glob(os.path.join(ROOT, "profiles", "r*_write_by_kernel.csv")))
    if not fetch or not write:
        return None, None

    def per_launch(path):
        tot, n = 0.0, 0
        for r in csv.DictReader(open(path)):
            if pat in r["kernel"]:
                tot += float(r["sum"]); n += int(r["launches"])
        return tot / n if n else None
    f, w = per_launch(fetch[-1]), per_launch(write[-1])
    if f is None or w is None:
        return None, None
    return int((2.0 * f + w) * 1024), f"{os.path.basename(fetch[-1])} + {os.path.basename(write[-1])} (committed rocprofv3 --pmc passes, Supervised-R50 workload)"


def host_cpu():
    """CPU model and the thread count SURVEY.md 8d prescribes: the physical cores of ONE socket, capped by what this process may use."""
    info = {}
    try:
        for line in subprocess.run(["lscpu"], capture_output=True, text=True, timeout=10).stdout.splitlines():
            k, _, v = line.partition(":")
            info[k.strip()] = v.strip()
    except Exception:
        pass
    def num(key, default):
        try:
            return int(info.get(key, default))
        except ValueError:
            return default
    per_socket = num("Core(s) per socket", 0)
    allowed = len(os.sched_getaffinity(0)) if hasattr(os, "sched_getaffinity") else (os.cpu_count() or 1)
    try:                                            # cgroup v2 CPU quota of the container, if any
        quota, period = open("/sys/fs/cgroup/cpu.max").read().split()
        if quota != "max":
            allowed = min(allowed, max(1, int(int(quota) / int(period))))
    except Exception:
        pass
    threads = max(1, min(per_socket or allowed, allowed))
    return {"cpu_model": info.get("Model name", "unknown"), "sockets": num("Socket(s)", 0), "cores_per_socket": per_socket, "usable_cpus": allowed}, threads


def cpu_baseline(args):
    """The CPU oracle (kind "port": our torch-CPU restatement, pinned to the reference by tests/golden) on a bounded sample:
    SURVEY.md 8d protocol -- threads = physical cores of one socket, 3 warm-up + 10 timed fwd+bwd+optimizer steps."""
    from oracle import models as OM
    wl = WORKLOADS[args.workload]
    enc = int(wl["enc"])
    B, H, W = args.cpu_batch, args.height, args.width
    host, threads = host_cpu()
    if args.cpu_threads > 0:
        threads = args.cpu_threads
    torch.set_num_threads(threads)
    threads = torch.get_num_threads()
    if wl.get("packnet"):
        sd, enc = OM.init_packnet_state_dict(wl["packnet"][-1], seed=0), "packnet" + wl["packnet"]
    else:
        sd = OM.init_state_dict(enc, with_pose=wl["arch"] != "SupDepthModel", seed=0)
    leaves = {k: v.clone().requires_grad_(True) for k, v in sd.items() if v.is_floating_point() and "running" not in k and "pixel" not in k and ".fc." not in k}
    state = dict(sd); state.update(leaves)
    opt = torch.optim.AdamW(list(leaves.values()), lr=1e-4, eps=1e-6) if wl["arch"] == "SupDepthModel" else torch.optim.Adam(list(leaves.values()), lr=2e-4)
    batch = synth_batch(wl["arch"], B, H, W, 1, "cpu")

    def step():
        opt.zero_grad()
        if wl["arch"] == "SupDepthModel":
            out = OM.supervised_forward(state, batch, enc, update_running=True)
            loss = out["silog_loss"]
        else:
            out = OM.monodepth2_forward(state, batch, enc, update_running=True, **({"var_w": 1e-4} if wl.get("packnet") else {}))
            loss = out["rec_loss"] + out["smooth_loss"] + out.get("var_loss", 0.0)
        loss.backward()
        opt.step()
    for _ in range(args.cpu_warmup):
        step()
    t0 = time.perf_counter()
    n = 0
    while n < args.cpu_steps:
        step(); n += 1
    dt = time.perf_counter() - t0
    out = {"value": round(B * n / dt, 3), "unit": "images/s", "cores": threads, "kind": "port",
           "sample": f"{n} timed fp32 training steps (fwd+bwd+optimizer) of the CPU oracle at bs={B}, {H}x{W}, after {args.cpu_warmup} warm-up steps; "
                     f"{round(dt / n * 1e3, 1)} ms per step"}
    out.update(host)
    return out


def self_launch(args):
    """`python bench.py --gpus N` without a launcher (the shape of the reference's `--num-gpus N`, detectron2/engine/launch.py:L24-92): the
    parent -- which has not touched the GPU -- starts N fresh rank processes with the torch.distributed environment, relays rank 0's JSON
    line and fails if any rank does."""
    import socket
    with socket.socket() as sk:
        sk.bind(("127.0.0.1", 0))
        port = sk.getsockname()[1]
    procs = []
    for r in range(args.gpus):
        env = dict(os.environ, RANK=str(r), LOCAL_RANK=str(r), WORLD_SIZE=str(args.gpus), LOCAL_WORLD_SIZE=str(args.gpus),
                   MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), HSA_ENABLE_IPC_MODE_LEGACY=os.environ.get("HSA_ENABLE_IPC_MODE_LEGACY", "0"))
        procs.append(subprocess.Popen([sys.executable, os.path.abspath(__file__)] + sys.argv[1:], env=env,
                                      stdout=subprocess.PIPE if r == 0 else subprocess.DEVNULL, text=True))
    # poll every rank: the first non-zero exit ends the run (the surviving ranks would otherwise sit in a collective until an outer timeout)
    import threading
    buf = []
    rd = threading.Thread(target=lambda: buf.append(procs[0].stdout.read()), daemon=True)
    rd.start()
    failed = None
    while failed is None and any(p.poll() is None for p in procs):
        for r, p in enumerate(procs):
            c = p.poll()
            if c is not None and c != 0:
                failed = (r, c)
                break
        time.sleep(0.2)
    if failed is None:
        failed = next(((r, p.returncode) for r, p in enumerate(procs) if p.returncode != 0), None)
    if failed is not None:
        for p in procs:
            if p.poll() is None:
                p.terminate()
        for p in procs:
            try:
                p.wait(timeout=10)
            except subprocess.TimeoutExpired:
                p.kill()
    rd.join(timeout=10)
    sys.stdout.write((buf[0] if buf else "") or "")
    sys.stdout.flush()
    if failed is not None:
        sys.stderr.write(f"bench.py: rank {failed[0]} exited with code {failed[1]}; the other ranks were terminated\n")
        sys.exit(failed[1] if 0 < failed[1] < 256 else 1)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=5)
    ap.add_argument("--workload", default="sup_r50", choices=sorted(WORKLOADS))
    ap.add_argument("--dtype", default="bf16", choices=["bf16", "fp32", "fp16"], help="fp16 = fp16 storage + dynamic loss scaling (SOLVER.AMP)")
    ap.add_argument("--batch", type=int, default=12)
    ap.add_argument("--height", type=int, default=192)
    ap.add_argument("--width", type=int, default=640)
    ap.add_argument("--no-graph", action="store_true", help="launch every kernel eagerly instead of replaying the captured hipGraph")
    ap.add_argument("--profile-steps", type=int, default=5, help="instrumented eager steps for the roofline object (0 = skip; at least 4 are run)")
    ap.add_argument("--cpu-steps", type=int, default=10)
    ap.add_argument("--cpu-warmup", type=int, default=3)
    ap.add_argument("--cpu-threads", type=int, default=0, help="0 = physical cores of one socket (SURVEY.md 8d)")
    ap.add_argument("--cpu-batch", type=int, default=2)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--with-loader", action="store_true", help="feed every step from pinned uint8 host frames through DevicePrefetcher + the device resize / "
                    "colour-jitter kernels (the input side inside the timed region) instead of device-resident inputs")
    ap.add_argument("--aug-on-copy-stream", action="store_true", help="--with-loader: run the resize / jitter kernels on the prefetcher's copy stream (then copied "
                    "into the graph's static inputs) instead of at the head of the step writing them in place (A/B)")
    ap.add_argument("--force-overlap", action="store_true", help="use the two-phase backward (all-reduce overlap path) even on one GPU")
    ap.add_argument("--no-side-stream", action="store_true", help="weight-gradient GEMMs on the main stream (single-stream graph: profiling aid)")
    ap.add_argument("--no-pose-stream", action="store_true", help="MonoDepth2: PoseNet on the main stream, after the depth network (A/B aid)")
    ap.add_argument("--marks", action="store_true", help="diagnostic: timestamp markers captured into the step's graph (sde_mark_time); the last "
                                                         "step's timeline is added to the line as marks_us (a few extra 1-thread launches per step)")
    ap.add_argument("--const", action="append", default=[], metavar="NAME=INT", help="A/B aid: scheduling constant of hip/lib.py (JOIN_LAG, WGRAD_GROUP, DEFER_MAX_BYTES, FORK_MIN_BYTES ...)")
    ap.add_argument("--opt", action="append", default=[], metavar="KEY=VALUE", help="sde_conv_set_option(KEY, VALUE) before the model is built (A/B measurements)")
    ap.add_argument("--no-pgemm", action="store_true", help="register-staged GEMM kernels only (A/B against the persistent LDS-DMA GEMM)")
    args = ap.parse_args()

    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        return self_launch(args)                    # before anything in this process touches the GPU
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    ndev = max(1, torch.cuda.device_count())
    dev_index = local_rank % ndev            # ranks > devices only in the single-GPU rehearsal below
    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        torch.cuda.set_device(dev_index)
        backend = os.environ.get("SDE_DIST_BACKEND", "nccl")      # "nccl" IS RCCL on ROCm; "gloo" lets 2 ranks rehearse on ONE GPU
        if backend == "nccl":
            dist.init_process_group("nccl", device_id=torch.device("cuda", dev_index))
        else:
            dist.init_process_group(backend)
    assert world == args.gpus, f"--gpus {args.gpus} but WORLD_SIZE={world}: launch with torch.distributed.run --nproc-per-node {args.gpus}"
    assert torch.cuda.is_available(), "bench.py needs a GPU (the HIP path has no CPU fallback)"
    device = torch.device("cuda", dev_index)
    torch.cuda.set_device(device)
    if os.environ.get("SDE_BENCH_STACKS"):          # debugging aid: every N seconds the Python stacks of all threads on stderr (where does a slow step sit?)
        import faulthandler
        faulthandler.dump_traceback_later(float(os.environ["SDE_BENCH_STACKS"]), repeat=True)

    if args.no_side_stream or args.no_pgemm or args.opt or args.const:
        from simpledepthestimation_amd.hip import lib as L, nn as HN
        for kv in args.const:
            k, v = kv.split("=")
            assert hasattr(L, k), k
            setattr(L, k, int(v))
            if k in ("JOIN_LAG", "WGRAD_GROUP", "FIRST_GROUP"):
                L.SCHEDULE_LOCKED = True
        for kv in args.opt:                     # A/B aid: --opt 6=128 -> sde_conv_set_option(SDE_OPT_WGRAD_BLOCKS, 128)
            k, v = kv.split("=")
            if k == "bn_fuse":                  # --opt bn_fuse=2 -> sde_bn_set_fuse(2)
                L.lib().sde_bn_set_fuse(int(v))
            elif k == "bnbwd":                  # --opt bnbwd=0 -> BatchNorm's backward reduce as its own pass everywhere (A/B)
                HN.BNBWD_FUSED = bool(int(v))
            elif k == "photo_multi":            # --opt photo_multi=0 -> one photometric launch per scale (A/B)
                from simpledepthestimation_amd.modeling.meta_arch import MonoDepth2 as MD
                MD.MULTI_SCALE_PHOTO = bool(int(v))
            elif k == "bias_defer":             # --opt bias_defer=0 -> every convolution's bias-gradient finalize as its own launch (A/B)
                HN.BIAS_DEFER = bool(int(v))
            elif k == "resbn":                  # --opt resbn=0 -> residual BatchNorms keep their own backward reduce pass (A/B)
                HN.RESBN_FUSED = bool(int(v))
            elif k == "early_tail":             # --opt early_tail=0 -> the last weight-gradient group waits for the phase's flush (A/B)
                HN.EARLY_TAIL = bool(int(v))
            elif k == "head_bias":              # --opt head_bias=0 -> disparity-head bias gradients by the separate pass
                HN.HEAD_BIAS_FUSED = bool(int(v))
            else:
                HN.set_option(int(k), int(v))
                if int(k) == HN.OPT_WGRAD_BLOCKS:
                    L.WGRAD_BLOCKS_LOCKED = True
        if args.no_side_stream:
            L.SIDE_STREAM = False
        if args.no_pgemm:
            HN.set_option(HN.OPT_PGEMM, 0)
    if args.marks:
        from simpledepthestimation_amd.hip import lib as L
        L.marks_enable(device)
    cfg, model, trainer = build(args, device)
    batch = synth_batch(WORKLOADS[args.workload]["arch"], args.batch, args.height, args.width, 1000 + rank, device)

    h2d_bytes = 0
    if args.with_loader:
        # the input side inside the timed region: pinned uint8 host batches -> DevicePrefetcher (copy stream) -> sde_image_prep_u8 -> trainer.step
        from simpledepthestimation_amd.data import DevicePrefetcher
        from simpledepthestimation_amd.data.device_aug import DeviceImageAug
        arch = WORKLOADS[args.workload]["arch"]
        # the prefetcher uploads (copy stream, persistent slot buffers); the resize + jitter kernels run at the head of the step and write the graph's
        # static inputs in place (measured both ways: the kernels on the copy stream + device-to-device copies into the static inputs cost 8-15 %)
        aug = DeviceImageAug(device)
        if not args.aug_on_copy_stream:
            trainer.input_transform = aug
        # ONE loader / prefetcher iteration for warm-up and timed region: the pinned host buffers and the prefetcher's slot buffers are allocated (and
        # their first transfers paid) during the warm-up, as in a training run that has been going for a while
        n_warm = max(args.warmup, 36)           # (covers the prefetcher's copy-stream selection: 32 batches)
        gen, h2d_bytes = host_loader(arch, args.batch, args.height, args.width, 2000 + rank, n_warm + args.steps)
        prefetcher = DevicePrefetcher(gen, device, device_aug=(aug if args.aug_on_copy_stream else None))
        feed = iter(prefetcher)
        for _ in range(n_warm):
            losses = trainer.step(next(feed))
    else:
        for _ in range(args.warmup):
            losses = trainer.step(batch)
    torch.cuda.synchronize()
    if world > 1:
        dist.barrier()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    if args.with_loader:
        for hb in feed:
            losses = trainer.step(hb)
    else:
        for _ in range(args.steps):
            losses = trainer.step(batch)
    torch.cuda.synchronize()
    if world > 1:
        dist.barrier()
    torch.cuda.synchronize()
    elapsed = time.perf_counter() - t0
    if world > 1:
        t = torch.tensor([elapsed], device=device, dtype=torch.float64)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = float(t.item())
    final = {k: float(v.item()) for k, v in losses.items()}
    assert all(x == x and abs(x) != float("inf") for x in final.values()), f"non-finite loss {final}"

    if rank == 0:
        ms = elapsed / args.steps * 1e3
        value = args.batch * world * args.steps / elapsed
        out = {"metric": "training images/sec at 192x640 bs=12/GPU", "value": round(value, 2), "unit": "images/s", "n_gpus": world, "steps": args.steps,
               "warmup": args.warmup, "ms_per_step": round(ms, 3), "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
               "dtype": {"bf16": "bf16", "fp16": "f16", "fp32": "f32"}[args.dtype], "data": "synthetic",
               "config": {"workload": f"{WORKLOADS[args.workload]['desc']}, {args.dtype} storage / fp32 accumulate, bs={args.batch}/GPU, "
                                      f"{args.height}x{args.width}, fwd+bwd+optimizer, random-init weights", "global_batch": args.batch * world,
                          "parallelism": f"dp{world}", "hip_graph": not args.no_graph, "allreduce_overlap": bool(trainer.overlap)},
               # the collective backend that actually ran ("rccl" = torch.distributed "nccl" on ROCm; "gloo" = the one-GPU rehearsal) and the
               # number of physical devices the ranks were spread over
               "backend": ("none" if world == 1 else {"nccl": "rccl"}.get(dist.get_backend(), dist.get_backend())),
               "rccl_ranks": (world if world > 1 and dist.get_backend() == "nccl" else 0), "devices": min(world, ndev),
               "final_losses": final}
        if args.marks:
            from simpledepthestimation_amd.hip import lib as L
            out["marks_us"] = {k: round(v, 1) for k, v in sorted(L.marks_read("step_start").items(), key=lambda kv: kv[1])}
        if args.with_loader:
            out["data"] = "synthetic uint8 frames at 375x1242 fed per step through the pinned host -> device prefetcher and the device resize + colour-jitter kernels"
            out["input_side"] = {"h2d_bytes_per_step": h2d_bytes, "in_timed_region": True, "copy_stream": getattr(prefetcher, "picked", None)}
        if args.profile_steps > 0:
            hbm, out["roofline"] = roofline_pass(trainer, batch, args.profile_steps, args.dtype, args.workload, step_ms=ms)
            if hbm is not None:
                out["roofline_photometric"] = hbm        # MonoDepth2 workloads: the HBM-bound warp+SSIM kernel next to the dominant GEMM
        if not args.no_cpu_baseline and world == 1:          # reported at N=1 only
            out["cpu_baseline"] = cpu_baseline(args)
        print(json.dumps(out), flush=True)
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
