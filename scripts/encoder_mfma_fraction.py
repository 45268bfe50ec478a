"""Encoder-only MFMA fraction: ResNet-50 encoder forward + backward at bs=12, 192x640, bf16, eager and serial.

    run under rocprofv3:   rocprofv3 --kernel-trace --output-format csv -d DIR -o TAG -- python3 scripts/encoder_mfma_fraction.py run
    then:                  python3 scripts/encoder_mfma_fraction.py report DIR/TAG_kernel_trace.csv   (scripts/gpu_encoder_fraction.sh does both)

Every kernel of the trace belongs to the encoder, so "encoder-only kernel time from rocprof" needs no attribution.  FLOPs: SURVEY 8d, 20.02 GFLOP
forward per image (3x3 9.06, 1x1 10.38, stem 0.58), training = 3x forward minus the data gradient of the stem (its input is the image).
"""
import collections
import csv
import os
import re
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
STEPS, B, H, W = 6, 12, 192, 640
FWD_GFLOP_PER_IMAGE, STEM_GFLOP = 20.02, 0.58
PEAK_TFLOPS = 2500.0


def run():
    import torch
    from simpledepthestimation_amd.hip import lib as L
    from simpledepthestimation_amd.layers.resnet_encoder import ResnetEncoder
    L.SIDE_STREAM = False                        # serial: kernel durations are not inflated by a co-running stream
    dev = "cuda"
    torch.manual_seed(0)
    enc = ResnetEncoder(num_layers=50, pretrained=False).to(dev).train()
    x = torch.randn(B, H, W, 8, device=dev).to(torch.bfloat16)       # NHWC, 3 channels padded to 8 (what sde_prep_input produces)
    x[..., 3:] = 0
    for _ in range(STEPS):
        feats = enc(x)
        loss = sum(f.float().square().mean() for f in feats)
        loss.backward()
        for p in enc.parameters():
            p.grad = None
    torch.cuda.synchronize()
    print("done", float(loss))


def report(path):
    rows = list(csv.DictReader(open(path)))
    fam = collections.defaultdict(lambda: [0, 0.0])
    gemm = re.compile(r"pgemm_kernel|halo3_kernel|igemm_kernel|wgrad_kernel|wgrad_dma_kernel|whalo_kernel|chalo_kernel")
    tot_gemm = tot_all = 0.0
    for r in rows:
        n = r["Kernel_Name"]
        d = (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3
        k = re.sub(r"\\(anonymous namespace\\)::", "", n)
        k = re.sub(r"^void ", "", k)
        k = re.sub(r"^_ZN\\d+_GLOBAL__N_1\\d+|^_ZN7sdeconv\\d+", "", k)[:60]
        fam[k][0] += 1; fam[k][1] += d
        tot_all += d
        if gemm.search(n):
            tot_gemm += d
    flops = (3 * FWD_GFLOP_PER_IMAGE - STEM_GFLOP) * B * 1e9
    per_step_gemm, per_step_all = tot_gemm / STEPS, tot_all / STEPS
    print(f"ResNet-50 encoder, bs={B}, {H}x{W}, bf16, fwd + bwd, {STEPS} eager steps (serial, no side stream)")
    print(f"algorithmic FLOPs per step: {flops / 1e9:.1f} GFLOP")
    print(f"GEMM kernels (pgemm / halo3 / igemm / wgrad): {per_step_gemm / 1e3:.3f} ms per step -> {flops / (per_step_gemm * 1e-6) / 1e12:.1f} TFLOP/s"
          f" = {100 * flops / (per_step_gemm * 1e-6) / 1e12 / PEAK_TFLOPS:.1f} % of the {PEAK_TFLOPS:.0f} TFLOP/s dense bf16 peak")
    print(f"all encoder kernels (GEMM + BatchNorm + weight-gradient reduction + pooling): {per_step_all / 1e3:.3f} ms per step -> "
          f"{flops / (per_step_all * 1e-6) / 1e12:.1f} TFLOP/s = {100 * flops / (per_step_all * 1e-6) / 1e12 / PEAK_TFLOPS:.1f} %")
    print(f"{'kernel':60s} {'launches/step':>13s} {'us/step':>9s}")
    for k, (n, t) in sorted(fam.items(), key=lambda kv: -kv[1][1])[:16]:
        print(f"{k:60s} {n / STEPS:13.1f} {t / STEPS:9.1f}")


if __name__ == "__main__":
    if len(sys.argv) > 1 and sys.argv[1] == "report":
        report(sys.argv[2])
    else:
        run()
