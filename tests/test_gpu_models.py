"""GPU parity: whole models built through build_model(cfg) (HIP path) vs the reference goldens and the CPU oracle.

north_star tolerance: depth maps within 1e-4 relative in fp32.  Gradients are compared per parameter tensor by relative L2.
"""
import os
import shutil

import pytest
import torch

from oracle import models as OM
from oracle.gen_golden import mono_batch, sup_batch

pytestmark = pytest.mark.gpu
dev = "cuda"


def make_cfg(arch, enc, dtype="fp32"):
    from simpledepthestimation_amd.config import get_project_cfg
    cfg = get_project_cfg("Supervised" if arch == "SupDepthModel" else "MonoDepth2")
    cfg.MODEL.META_ARCHITECTURE = arch
    if str(enc).startswith("packnet"):           # "packnet1A" / "packnet1B": projects/MonoDepth2/configs/packnet_1a.yaml
        cfg.MODEL.DEPTH_NET.NAME = "PackNet01"
        cfg.MODEL.DEPTH_NET.VERSION = str(enc)[-2:]
        cfg.LOSS.VAR_LOSS_WEIGHT = 1e-4
        enc = 18
    cfg.MODEL.DEPTH_NET.ENCODER_NAME = str(enc)
    cfg.MODEL.COMPUTE_DTYPE = dtype
    cfg.MODEL.DEVICE = dev
    return cfg


def build(arch, enc, sd, dtype="fp32"):
    from simpledepthestimation_amd.modeling import build_model
    m = build_model(make_cfg(arch, enc, dtype))
    m.load_state_dict(sd, strict=True)
    return m


def rel(a, b):
    a, b = a.detach().double().cpu(), b.detach().double().cpu()
    return ((a - b).norm() / (b.norm() + 1e-30)).item()


def max_rel(a, b):
    a, b = a.detach().double().cpu(), b.detach().double().cpu()
    return ((a - b).abs() / b.abs().clamp(min=1e-12)).max().item()


def clone_batch(b):
    return {k: ([x.clone() for x in v] if isinstance(v, list) else (v.clone() if torch.is_tensor(v) else v)) for k, v in b.items()}


@pytest.mark.parametrize("tag,enc,B", [("sup18", 18, 2), ("sup50", 50, 1)])
def test_supervised_vs_reference_golden(mod, tag, enc, B):
    """BASELINE.json config 1 (R18, bs=2, 64x192) and an R50 case: losses/depths/grad norms produced by the REFERENCE."""
    sd = OM.init_state_dict(enc, seed=100 + enc)
    model = build("SupDepthModel", enc, sd).train()
    batch = sup_batch(B, 64, 192, 3)
    out = model(clone_batch(batch))
    assert abs(out["silog_loss"].item() - float(mod[f"{tag}.silog_loss"])) < 5e-5 * float(mod[f"{tag}.silog_loss"])
    for i in range(4):
        assert out["depth_pred"][i].shape == mod.t(f"{tag}.depth{i}").shape
        assert max_rel(out["depth_pred"][i], mod.t(f"{tag}.depth{i}")) < 1e-4, f"depth scale {i}"
    out["silog_loss"].backward()
    named = dict(model.named_parameters())
    for k in [k for k in mod.keys() if k.startswith(f"{tag}.gnorm.")]:
        n = k[len(tag) + 7:]
        g = named[n].grad.norm().item()
        assert abs(g - float(mod[k])) < 3e-3 * float(mod[k]) + 1e-7, f"grad norm of {n}: {g} vs {float(mod[k])}"
    e = model.depth_net.encoder.encoder
    assert rel(e.bn1.running_mean, mod.t(f"{tag}.bn1_running_mean")) < 1e-5
    assert rel(e.bn1.running_var, mod.t(f"{tag}.bn1_running_var")) < 1e-5
    assert int(model.state_dict()["depth_net.encoder.encoder.bn1.num_batches_tracked"]) == 1
    # flip branch (DepthResNet.py:L52-60), still train mode
    fb = clone_batch(batch); fb["flip"] = True
    with torch.no_grad():
        fl = model(fb)
    assert max_rel(fl["depth_pred"][0], mod.t(f"{tag}.flip_depth0")) < 1e-4
    # eval mode uses the running statistics; the golden was taken after exactly one training forward
    m2 = build("SupDepthModel", enc, sd).train()
    with torch.no_grad():
        m2(clone_batch(batch))
    m2.eval()
    with torch.no_grad():
        ev = m2(clone_batch(batch))
    assert ev["depth_pred"].shape == (B, 1, 64, 192)
    assert max_rel(ev["depth_pred"], mod.t(f"{tag}.eval_depth")) < 1e-4


def test_supervised_all_gradients_vs_oracle():
    """Every parameter gradient of the R18 supervised step vs the CPU oracle's autograd."""
    sd = OM.init_state_dict(18, seed=5)
    batch = sup_batch(2, 64, 192, 8)
    sdo = {k: (v.clone().requires_grad_(True) if v.is_floating_point() and "running" not in k and "pixel" not in k else v.clone())
           for k, v in sd.items()}
    OM.supervised_forward(sdo, batch, 18)["silog_loss"].backward()
    model = build("SupDepthModel", 18, sd).train()
    model(clone_batch(batch))["silog_loss"].backward()
    worst = ("", 0.0)
    for n, p in model.named_parameters():
        if ".fc." in n:
            assert p.grad is None
            continue
        e = rel(p.grad, sdo[n].grad)
        if e > worst[1]:
            worst = (n, e)
        assert e < 2e-3, f"{n}: rel grad error {e:.3e}"
    print("worst parameter-gradient error:", worst)


@pytest.mark.parametrize("tag,H,W", [("mono18", 64, 192), ("mono18_full", 192, 640)])
def test_monodepth2_vs_reference_golden(mod, tag, H, W):
    sd = OM.init_state_dict(18, with_pose=True, seed=7)
    model = build("MonoDepth2Model", 18, sd).train()
    batch = mono_batch(2, H, W, 21)
    out = model(clone_batch(batch))
    assert set(k for k in out if "loss" in k) == {"rec_loss", "smooth_loss"}
    assert abs(out["rec_loss"].item() - float(mod[f"{tag}.rec_loss"])) < 5e-5 * float(mod[f"{tag}.rec_loss"])
    assert abs(out["smooth_loss"].item() - float(mod[f"{tag}.smooth_loss"])) < 5e-4 * float(mod[f"{tag}.smooth_loss"])
    (out["rec_loss"] + out["smooth_loss"]).backward()
    named = dict(model.named_parameters())
    for k in [k for k in mod.keys() if k.startswith(f"{tag}.gnorm.")]:
        n = k[len(tag) + 7:]
        g = named[n].grad.norm().item()
        assert abs(g - float(mod[k])) < 1e-2 * float(mod[k]) + 1e-8, f"grad norm of {n}: {g} vs {float(mod[k])}"
    if tag == "mono18":
        with torch.no_grad():
            b2 = clone_batch(batch)
            b2 = {k: ([x.to(dev) for x in v] if isinstance(v, list) else v.to(dev)) for k, v in b2.items()}
            b2["pose_net_input"] = torch.cat([b2["img"]] + b2["ctx_img"], 1)
            poses = model.pose_net(b2)["pose_pred"]
        assert rel(poses[0], mod.t("mono18.pose0")) < 1e-5 and rel(poses[1], mod.t("mono18.pose1")) < 1e-5


def test_monodepth2_resnet50_vs_reference_golden(mono50):
    """BASELINE.json configs[3] (projects/MonoDepth2/configs/resnet50.yaml): MonoDepth2Model + ResNet-50 encoder through build_model on the HIP
    path (fp32) vs losses / gradient norms / eval-mode depth produced by the REFERENCE (tests/golden/mono50.npz, oracle/gen_golden.py)."""
    tag = "mono50"
    sd = OM.init_state_dict(50, with_pose=True, seed=57)
    model = build("MonoDepth2Model", 50, sd).train()
    batch = mono_batch(1, 64, 192, 23)
    out = model(clone_batch(batch))
    assert set(k for k in out if "loss" in k) == {"rec_loss", "smooth_loss"}
    assert abs(out["rec_loss"].item() - float(mono50[f"{tag}.rec_loss"])) < 5e-5 * float(mono50[f"{tag}.rec_loss"])
    assert abs(out["smooth_loss"].item() - float(mono50[f"{tag}.smooth_loss"])) < 5e-4 * float(mono50[f"{tag}.smooth_loss"])
    (out["rec_loss"] + out["smooth_loss"]).backward()
    named = dict(model.named_parameters())
    for k in [k for k in mono50.keys() if k.startswith(f"{tag}.gnorm.")]:
        n = k[len(tag) + 7:]
        g = named[n].grad.norm().item()
        assert abs(g - float(mono50[k])) < 1e-2 * float(mono50[k]) + 1e-8, f"grad norm of {n}: {g} vs {float(mono50[k])}"
    model.eval()
    with torch.no_grad():
        d = model(clone_batch(batch))["depth_pred"]
    assert max_rel(d, mono50.t(f"{tag}.eval_depth")) < 1e-4        # north_star: depth maps within 1e-4 relative (fp32)


def test_monodepth2_eval_and_options():
    sd = OM.init_state_dict(18, with_pose=True, seed=7)
    model = build("MonoDepth2Model", 18, sd).eval()
    batch = mono_batch(2, 64, 192, 21)
    with torch.no_grad():
        out = model(clone_batch(batch))
    assert list(out.keys()) == ["depth_pred"] and out["depth_pred"].shape == (2, 1, 64, 192)
    from simpledepthestimation_amd.modeling import build_model
    cfg = make_cfg("MonoDepth2Model", 18)
    cfg.LOSS.AUTOMASK = False
    cfg.LOSS.PHOTOMETRIC_REDUCE = "mean"
    m = build_model(cfg)
    m.load_state_dict(sd)
    m.train()
    out = m(clone_batch(batch))
    sdo = {k: v.clone() for k, v in sd.items()}
    ref = OM.monodepth2_forward(sdo, batch, 18, automask=False, reduce="mean")
    assert abs(out["rec_loss"].item() - ref["rec_loss"].item()) < 5e-5 * ref["rec_loss"].item()


@pytest.mark.parametrize("enc", [18, 50])
def test_supervised_bf16_tracks_fp32(enc):
    """bf16 throughput mode (BASELINE config 2): same weights, loss and depth close to the fp32 path, gradients aligned.

    Gradients are compared by direction: tightly where the path from the loss is short (decoder), and globally over the
    concatenated gradient; deep encoder layers accumulate bf16 rounding through up to 50 BatchNorm'd layers.
    """
    sd = OM.init_state_dict(enc, seed=3)
    batch = sup_batch(4, 128, 416, 4)
    m32 = build("SupDepthModel", enc, sd, "fp32").train()
    m16 = build("SupDepthModel", enc, sd, "bf16").train()
    o32 = m32(clone_batch(batch)); o16 = m16(clone_batch(batch))
    assert abs(o16["silog_loss"].item() - o32["silog_loss"].item()) < 3e-2 * o32["silog_loss"].item()
    assert rel(o16["depth_pred"][0], o32["depth_pred"][0]) < 3e-2
    o32["silog_loss"].backward(); o16["silog_loss"].backward()
    cos_min, all32, all16 = (1.0, ""), [], []
    for (n, p32), (_, p16) in zip(m32.named_parameters(), m16.named_parameters()):
        if p32.grad is None:
            continue
        a, b = p32.grad.flatten().double(), p16.grad.flatten().double()
        assert torch.isfinite(b).all(), n
        all32.append(a); all16.append(b)
        if a.norm() < 1e-12:
            continue
        cos = float((a @ b) / (a.norm() * b.norm() + 1e-30))
        if cos < cos_min[0]:
            cos_min = (cos, n)
        if ".decoder." in n:
            assert cos > 0.97, f"{n}: bf16 decoder gradient direction diverges from fp32 (cos {cos:.3f})"
    a, b = torch.cat(all32), torch.cat(all16)
    total = float((a @ b) / (a.norm() * b.norm()))
    print(f"R{enc}: cosine(bf16 grad, fp32 grad) over all parameters = {total:.4f}; worst tensor {cos_min}")
    assert total > 0.95 and cos_min[0] > 0.0, (total, cos_min)


def test_registry_and_plugin_surface():
    from simpledepthestimation_amd.modeling import DEPTH_NET_REGISTRY, META_ARCH_REGISTRY, POSE_NET_REGISTRY
    from simpledepthestimation_amd.layers.fakeDDP import FakeDDP
    assert "SupDepthModel" in META_ARCH_REGISTRY and "MonoDepth2Model" in META_ARCH_REGISTRY
    assert "DepthResNet" in DEPTH_NET_REGISTRY and "PoseNet" in POSE_NET_REGISTRY
    with pytest.raises(KeyError):
        META_ARCH_REGISTRY.get("NoSuchModel")
    sd = OM.init_state_dict(18, with_pose=True, seed=7)
    model = FakeDDP(build("MonoDepth2Model", 18, sd))
    # attribute tree the project loops rely on (projects/*/train.py)
    assert model.module.depth_net.encoder is not None and model.module.depth_net.decoder is not None and model.module.pose_net is not None
    assert model.module.device.type == "cuda"


def test_two_phase_backward_equals_plain_backward():
    """engine.trainer's overlap path (autograd graph cut at the encoder features) must give bit-identical gradients and updates."""
    from simpledepthestimation_amd.engine.trainer import supervised_trainer
    sd = OM.init_state_dict(18, seed=11)
    batch = sup_batch(2, 64, 192, 12)
    dbatch = {k: v.to(dev) for k, v in batch.items()}
    res = []
    for overlap in (False, True):
        model = build("SupDepthModel", 18, sd).train()
        tr = supervised_trainer(model, make_cfg("SupDepthModel", 18), overlap=overlap)
        assert (tr._cut is not None) == overlap
        for _ in range(2):
            tr.step(clone_batch(dbatch))
        res.append((tr.gflat.clone(), tr.pflat.clone()))
    assert torch.equal(res[0][0], res[1][0]), "gradients differ between plain and two-phase backward"
    assert torch.equal(res[0][1], res[1][1])
    assert res[1][0].abs().sum() > 0


@pytest.mark.parametrize("use_graph", [False, True], ids=["eager", "graph"])
def test_two_phase_backward_with_posenet_on_the_auxiliary_stream_equals_the_plain_step(use_graph):
    """The N > 1 step shape of MonoDepth2 (autograd graph cut at the encoder features, phase A / phase B, PoseNet on the auxiliary stream
    underneath phase A) against the plain single-stream, single-phase step: losses, gradients and parameters bit for bit."""
    from simpledepthestimation_amd.engine import trainer as T
    sd = OM.init_state_dict(18, with_pose=True, seed=11)
    batch = mono_batch(2, 64, 192, 12)
    dbatch = {k: ([x.to(dev) for x in v] if isinstance(v, list) else v.to(dev) if torch.is_tensor(v) else v) for k, v in batch.items()}
    res = []
    for overlap, pose_stream in ((False, False), (True, True)):
        model = build("MonoDepth2Model", 18, sd).train()
        tr = T.monodepth2_trainer(model, make_cfg("MonoDepth2Model", 18), use_graph=use_graph, overlap=overlap, pose_stream=pose_stream)
        assert (tr._cut is not None) == overlap and tr.pose_stream == pose_stream
        losses = []
        for _ in range(3):
            out = tr.step(clone_batch(dbatch))
            losses.append({k: float(v) for k, v in out.items()})
        torch.cuda.synchronize()
        res.append((losses, tr.gflat.clone(), tr.pflat.clone()))
    assert res[0][0] == res[1][0], (res[0][0], res[1][0])
    assert torch.equal(res[0][1], res[1][1]), "gradients differ"
    assert torch.equal(res[0][2], res[1][2])


@pytest.mark.parametrize("arch", ["SupDepthModel", "MonoDepth2Model"])
def test_deferred_wgrad_reduce_equals_immediate(arch):
    """One batched slab-reduction launch per backward phase (WGradReducer) == the per-layer reductions, bit for bit."""
    from simpledepthestimation_amd.engine import trainer as T
    sd = OM.init_state_dict(18, with_pose=arch != "SupDepthModel", seed=11)
    batch = sup_batch(2, 64, 192, 12) if arch == "SupDepthModel" else mono_batch(2, 64, 192, 12)
    dbatch = {k: ([x.to(dev) for x in v] if isinstance(v, list) else v.to(dev) if torch.is_tensor(v) else v) for k, v in batch.items()}
    mk = T.supervised_trainer if arch == "SupDepthModel" else T.monodepth2_trainer
    res = []
    for deferred in (False, True):
        model = build(arch, 18, sd).train()
        tr = mk(model, make_cfg(arch, 18))
        if not deferred:
            tr._wreduce = None
        for _ in range(2):
            tr.step(clone_batch(dbatch))
        res.append((tr.gflat.clone(), tr.pflat.clone()))
    assert torch.equal(res[0][0], res[1][0]), "gradients differ between immediate and deferred weight-gradient reduction"
    assert torch.equal(res[0][1], res[1][1])
    assert res[1][0].abs().sum() > 0


@pytest.mark.parametrize("use_graph", [False, True], ids=["eager", "graph"])
def test_posenet_on_the_auxiliary_stream_equals_the_single_stream_step(use_graph):
    """MonoDepth2 with PoseNet on the auxiliary stream (underneath the depth network, forward and backward; HipTrainer's default for single-process
    training) against the same steps with everything on one stream: losses, gradients and parameters bit for bit."""
    from simpledepthestimation_amd.engine import trainer as T
    sd = OM.init_state_dict(18, with_pose=True, seed=11)
    batch = mono_batch(2, 64, 192, 12)
    dbatch = {k: ([x.to(dev) for x in v] if isinstance(v, list) else v.to(dev) if torch.is_tensor(v) else v) for k, v in batch.items()}
    res = []
    for pose_stream in (False, True):
        model = build("MonoDepth2Model", 18, sd).train()
        tr = T.monodepth2_trainer(model, make_cfg("MonoDepth2Model", 18), use_graph=use_graph, pose_stream=pose_stream)
        assert tr.pose_stream == pose_stream
        losses = []
        for _ in range(3):
            out = tr.step(clone_batch(dbatch))
            losses.append({k: float(v) for k, v in out.items()})
        torch.cuda.synchronize()
        res.append((losses, tr.gflat.clone(), tr.pflat.clone()))
    assert res[0][0] == res[1][0], (res[0][0], res[1][0])
    assert torch.equal(res[0][1], res[1][1]), "gradients differ"
    assert torch.equal(res[0][2], res[1][2])
    assert res[1][1].abs().sum() > 0


def test_head_bias_gradient_from_the_depth_head_backward():
    """The disparity heads' bias gradients summed inside depth_head's backward (trainer path: straight into the flat gradient) against the separate
    activation-backward pass over the padded gradient tensor: the same summands in another order."""
    from simpledepthestimation_amd.engine.trainer import supervised_trainer
    from simpledepthestimation_amd.hip import nn as HN
    sd = OM.init_state_dict(18, seed=11)
    batch = {k: v.to(dev) for k, v in sup_batch(2, 64, 192, 12).items()}
    res = []
    for fused in (False, True):
        HN.HEAD_BIAS_FUSED = fused
        try:
            model = build("SupDepthModel", 18, sd, "bf16").train()
            tr = supervised_trainer(model, make_cfg("SupDepthModel", 18, "bf16"))
            hits = HN.HEAD_BIAS_HITS
            tr.step(clone_batch(batch))
            torch.cuda.synchronize()
            assert HN.HEAD_BIAS_HITS - hits == (4 if fused else 0), "the four disparity heads must take the fused path exactly when it is on"
            names = [n for n, p in model.named_parameters() if n.endswith("conv.bias") and p.numel() == 1]
            res.append((tr.gflat.clone(), {n: p.grad.clone() for n, p in model.named_parameters() if n in names}))
        finally:
            HN.HEAD_BIAS_FUSED = True
    assert len(res[1][1]) == 4, list(res[1][1])
    for n, g in res[1][1].items():
        assert torch.allclose(g, res[0][1][n], rtol=2e-5, atol=1e-7) and float(g.abs()) > 0, (n, g, res[0][1][n])
    others = (res[0][0] - res[1][0]).abs()
    assert int((others > 0).sum()) <= 4      # nothing but the four bias gradients may differ at all


def test_head_bias_fused_path_refuses_a_second_consumer_of_the_logit():
    """The fused disparity-head bias gradient owns the logit's gradient only when depth_head is the logit's ONLY consumer: with a second one autograd
    hands the convolution a summed tensor, and adding its full column sum on top of the head's share would double count -- that raises instead."""
    from simpledepthestimation_amd.hip import nn as HN
    g = torch.Generator().manual_seed(31)
    x = torch.randn(2, 8, 16, 16, generator=g).bfloat16().to(dev)
    w = (torch.randn(1, 16, 3, 3, generator=g) * 0.1).to(dev).requires_grad_(True)
    b = torch.zeros(1, device=dev, requires_grad=True)
    w.grad, b.grad = torch.zeros_like(w), torch.zeros_like(b)          # pre-allocated slots, as under HipTrainer

    def run(second):
        w.grad.zero_(); b.grad.zero_()
        y = HN.conv2d(x, w, b, stride=1, pad=1, reflect=True)
        d = HN.depth_head(y, 0.1, 80.0)
        loss = d.sum() + (y.float().sum() if second else 0.0)
        loss.backward()
        torch.cuda.synchronize()
        return b.grad.clone()
    hits = HN.HEAD_BIAS_HITS
    one = run(False)
    assert HN.HEAD_BIAS_HITS == hits + 1 and float(one.abs()) > 0
    with pytest.raises(Exception, match="second consumer"):
        run(True)
    HN.HEAD_BIAS_FUSED = False
    try:
        two = run(True)                                   # the separate pass handles any graph: head share + 1 per pixel from the second consumer
    finally:
        HN.HEAD_BIAS_FUSED = True
    assert abs(float(two) - float(one) - 2 * 8 * 16) < 1e-2 * abs(float(two))
    assert torch.allclose(run(False), one)               # nothing stale is left behind by the refused backward


def test_monodepth2_multi_scale_photometric_launch_equals_the_per_scale_launches():
    """MonoDepth2Model with every scale of the photometric + smoothness terms in one launch per phase (MULTI_SCALE_PHOTO, sde_mono_loss_*) against one
    launch per scale and term: the same losses, the same gradients up to the order in which the pose gradients of the scales are summed."""
    from simpledepthestimation_amd.modeling.meta_arch import MonoDepth2 as MD
    sd = OM.init_state_dict(18, with_pose=True, seed=7)
    batch = mono_batch(2, 64, 192, 21)
    res = []
    kept = MD.MULTI_SCALE_PHOTO
    for multi in (False, True):
        MD.MULTI_SCALE_PHOTO = multi
        try:
            model = build("MonoDepth2Model", 18, sd).train()
            out = model(clone_batch(batch))
            (out["rec_loss"] + out["smooth_loss"]).backward()
            torch.cuda.synchronize()
            res.append((float(out["rec_loss"]), float(out["smooth_loss"]), {n: p.grad.clone() for n, p in model.named_parameters() if p.grad is not None}))
        finally:
            MD.MULTI_SCALE_PHOTO = kept
    # per-scale terms are bit-identical (test_mono_loss_all_scales); the weighted totals are summed in scale order here and by a dot product there
    assert abs(res[0][0] - res[1][0]) <= 2e-7 * abs(res[0][0]) and abs(res[0][1] - res[1][1]) <= 2e-7 * abs(res[0][1])
    norms = sorted(float(g.norm()) for g in res[0][2].values())
    floor = 1e-3 * norms[len(norms) // 2]      # a conv bias in front of a GroupNorm has a zero gradient in exact arithmetic: rounding noise on both sides
    for n in res[0][2]:
        if n.startswith("pose_net.") and n.endswith(".0.bias"):
            continue        # conv bias in front of GroupNorm: exactly zero gradient, what is computed is cancellation noise of the (re-ordered) pose-gradient sums
        a, b = res[1][2][n].double(), res[0][2][n].double()
        assert float((a - b).norm() / (b.norm() + floor)) < 1e-4, n


def test_graph_replay_equals_eager():
    """The captured hipGraph step (zero-grad + batched weight pack + forward + backward) reproduces the eager step bit for bit."""
    from simpledepthestimation_amd.engine.trainer import supervised_trainer
    sd = OM.init_state_dict(18, seed=11)
    batch = {k: v.to(dev) for k, v in sup_batch(2, 64, 192, 12).items()}
    res = []
    for use_graph in (False, True):
        model = build("SupDepthModel", 18, sd).train()
        tr = supervised_trainer(model, make_cfg("SupDepthModel", 18), use_graph=use_graph)
        losses = [float(tr.step(clone_batch(batch))["silog_loss"]) for _ in range(3)]
        res.append((losses, tr.pflat.clone(), {k: v.clone() for k, v in model.state_dict().items()}))
    assert res[0][0] == res[1][0], (res[0][0], res[1][0])
    assert torch.equal(res[0][1], res[1][1])
    for k, v in res[0][2].items():               # buffers too: the capture's warm-up steps leave no trace in the BatchNorm running statistics
        assert torch.equal(v, res[1][2][k]), k
    assert int(res[1][2]["depth_net.encoder.encoder.bn1.num_batches_tracked"]) == 3


@pytest.mark.parametrize("version", ["A", "B"])
def test_packnet_vs_reference_golden(pack, version):
    """MonoDepth2Model + PackNet01 (BASELINE config 5 shape at 64x192): losses, probe gradient norms, depth maps and the flip branch
    against the reference's goldens; every parameter gradient against the CPU oracle."""
    tag = "packnet1" + version
    sd = OM.init_packnet_state_dict(version, seed=5)
    model = build("MonoDepth2Model", tag, sd).train()
    batch = mono_batch(1, 64, 192, 21)
    out = model(clone_batch(batch))
    assert set(k for k in out if "loss" in k) == {"rec_loss", "smooth_loss", "var_loss"}
    assert abs(out["rec_loss"].item() - float(pack[f"{tag}.rec_loss"])) < 1e-4 * float(pack[f"{tag}.rec_loss"])
    assert abs(out["smooth_loss"].item() - float(pack[f"{tag}.smooth_loss"])) < 2e-3 * float(pack[f"{tag}.smooth_loss"])
    assert abs(out["var_loss"].item() - float(pack[f"{tag}.var_loss"])) < 2e-3 * float(pack[f"{tag}.var_loss"])
    (out["rec_loss"] + out["smooth_loss"] + out["var_loss"]).backward()
    named = dict(model.named_parameters())
    for k in [k for k in pack.keys() if k.startswith(f"{tag}.gnorm.")]:
        n = k[len(tag) + 7:]
        g = named[n].grad.norm().item()
        assert abs(g - float(pack[k])) < 1e-2 * float(pack[k]) + 1e-8, f"grad norm of {n}: {g} vs {float(pack[k])}"
    # all gradients vs the oracle (fp32 on both sides)
    leaves = {k: v.clone().requires_grad_(True) for k, v in sd.items() if v.is_floating_point() and "pixel" not in k}
    state = dict(sd); state.update(leaves)
    o = OM.monodepth2_forward(state, batch, tag, var_w=1e-4)
    (o["rec_loss"] + o["smooth_loss"] + o["var_loss"]).backward()
    worst, bad = 0.0, []
    norms = sorted(float(v.grad.norm()) for v in leaves.values() if v.grad is not None)
    floor = 1e-2 * norms[len(norms) // 2]       # gradients that are zero in exact arithmetic (a conv bias in front of a GroupNorm with one
    for n, p in named.items():                  # channel per group) are rounding noise on both sides: measure errors against this floor
        ref = leaves[n].grad
        if ref is None or ref.numel() == 1:
            continue        # 1-element gradients (the InvDepth biases) are sums with heavy cancellation: fp32 order noise on both sides
        e = (p.grad.detach().double().cpu() - ref.double()).norm().item() / (ref.double().norm().item() + floor)
        worst = max(worst, e)
        if e >= 2e-2:
            bad.append(f"{n}: {e:.3e}")
    assert not bad, "gradient relative errors: " + "; ".join(bad)
    assert worst > 0
    model.eval()
    with torch.no_grad():
        b2 = {k: ([x.to(dev) for x in v] if isinstance(v, list) else v.to(dev)) for k, v in clone_batch(batch).items()}
        b2["depth_net_input"] = (b2["img"] - model.pixel_mean) / model.pixel_std
        depths = model.depth_net(dict(b2))["depth_pred"]
        for i, d in enumerate(depths):
            assert rel(d, pack.t(f"{tag}.depth{i}")) < 1e-4, f"depth{i}: {rel(d, pack.t(f'{tag}.depth{i}'))}"
        fb = dict(b2); fb["flip"] = True
        assert rel(model.depth_net(fb)["depth_pred"][0], pack.t(f"{tag}.flip_depth0")) < 1e-4


def test_packnet_bf16_step_runs():
    """bf16 PackNet01 training steps through the trainer (flat buffers, deferred reductions, Adam): finite losses, parameters move, and the
    bf16 loss stays within 3 % of the fp32 reference golden at step 0."""
    from simpledepthestimation_amd.engine.trainer import monodepth2_trainer
    sd = OM.init_packnet_state_dict("A", seed=5)
    model = build("MonoDepth2Model", "packnet1A", sd, "bf16").train()
    tr = monodepth2_trainer(model, make_cfg("MonoDepth2Model", "packnet1A", "bf16"))
    batch = {k: ([x.to(dev) for x in v] if isinstance(v, list) else v.to(dev)) for k, v in mono_batch(1, 64, 192, 21).items()}
    p0 = tr.pflat.clone()
    losses = [sum(float(v) for v in tr.step(clone_batch(batch)).values()) for _ in range(4)]
    assert all(x == x and abs(x) != float("inf") for x in losses), losses
    assert (tr.pflat != p0).float().mean() > 0.9
    from conftest import _Golden, GOLDEN
    import os
    pk = _Golden(os.path.join(GOLDEN, "packnet.npz"))
    ref0 = float(pk["packnet1A.rec_loss"]) + float(pk["packnet1A.smooth_loss"]) + float(pk["packnet1A.var_loss"])
    assert abs(losses[0] - ref0) < 3e-2 * ref0, (losses[0], ref0)


def test_monodepth2_upsample_depth_training(opt):
    """UPSAMPLE_DEPTH=True with gradients (nearest up-sampling of the four depth maps, losses at full resolution) vs the reference golden."""
    sd = OM.init_state_dict(18, with_pose=True, seed=7)
    from simpledepthestimation_amd.modeling import build_model
    cfg = make_cfg("MonoDepth2Model", 18)
    cfg.MODEL.DEPTH_NET.UPSAMPLE_DEPTH = True
    model = build_model(cfg)
    model.load_state_dict(sd, strict=True)
    model.train()
    out = model(clone_batch(mono_batch(2, 64, 192, 21)))
    assert abs(out["rec_loss"].item() - float(opt["mono18_up.rec_loss"])) < 5e-5 * float(opt["mono18_up.rec_loss"])
    assert abs(out["smooth_loss"].item() - float(opt["mono18_up.smooth_loss"])) < 5e-4 * float(opt["mono18_up.smooth_loss"])
    (out["rec_loss"] + out["smooth_loss"]).backward()
    named = dict(model.named_parameters())
    for k in [k for k in opt.keys() if k.startswith("mono18_up.gnorm.")]:
        n = k[len("mono18_up.gnorm."):]
        g = named[n].grad.norm().item()
        assert abs(g - float(opt[k])) < 1e-2 * float(opt[k]) + 1e-8, f"grad norm of {n}: {g} vs {float(opt[k])}"


@pytest.mark.parametrize("red", ["min", "mean"])
def test_monodepth2_loss_clip(opt, red):
    """LOSS.CLIP = 0.5 on the fused kernels (thresholds from one extra map pass, clipped pixels carry no gradient) vs the reference golden."""
    tag = f"mono18_clip_{red}"
    sd = OM.init_state_dict(18, with_pose=True, seed=7)
    from simpledepthestimation_amd.modeling import build_model
    cfg = make_cfg("MonoDepth2Model", 18)
    cfg.LOSS.CLIP = 0.5
    cfg.LOSS.PHOTOMETRIC_REDUCE = red
    model = build_model(cfg)
    model.load_state_dict(sd, strict=True)
    model.train()
    out = model(clone_batch(mono_batch(2, 64, 192, 21)))
    assert abs(out["rec_loss"].item() - float(opt[f"{tag}.rec_loss"])) < 1e-4 * float(opt[f"{tag}.rec_loss"])
    (out["rec_loss"] + out["smooth_loss"]).backward()
    named = dict(model.named_parameters())
    for k in [k for k in opt.keys() if k.startswith(f"{tag}.gnorm.")]:
        n = k[len(tag) + 7:]
        g = named[n].grad.norm().item()
        assert abs(g - float(opt[k])) < 2e-2 * float(opt[k]) + 1e-8, f"grad norm of {n}: {g} vs {float(opt[k])}"


def test_checkpoint_resume_continues_bit_exactly(tmp_path):
    """Save (model + fused-Adam state) after 2 steps, resume in a freshly built model/trainer: steps 3-4 reproduce the uninterrupted run bit
    for bit, graph replay included (the replayed graph reads the flat buffers the loader writes into)."""
    from simpledepthestimation_amd.checkpoint import DetectionCheckpointer, PeriodicCheckpointer
    from simpledepthestimation_amd.engine.trainer import monodepth2_trainer
    cfg = make_cfg("MonoDepth2Model", 18)
    batch = mono_batch(2, 64, 192, 12)
    dbatch = {k: ([x.to(dev) for x in v] if isinstance(v, list) else v.to(dev) if torch.is_tensor(v) else v) for k, v in batch.items()}
    model = build("MonoDepth2Model", 18, OM.init_state_dict(18, with_pose=True, seed=21)).train()
    tr = monodepth2_trainer(model, cfg, use_graph=True)
    for _ in range(2):
        tr.step(clone_batch(dbatch))
    PeriodicCheckpointer(DetectionCheckpointer(model, str(tmp_path), optimizer=tr), 1).step(1)
    want = [float(tr.step(clone_batch(dbatch))["rec_loss"]) for _ in range(2)]
    model2 = build("MonoDepth2Model", 18, OM.init_state_dict(18, with_pose=True, seed=99)).train()        # different weights on purpose
    tr2 = monodepth2_trainer(model2, cfg, use_graph=True)
    rest = DetectionCheckpointer(model2, str(tmp_path), optimizer=tr2).resume_or_load("", resume=True)
    assert rest == {"iteration": 1} and tr2.t == 2
    got = [float(tr2.step(clone_batch(dbatch))["rec_loss"]) for _ in range(2)]
    assert got == want, (got, want)
    assert torch.equal(tr.pflat, tr2.pflat) and torch.equal(tr.m, tr2.m) and torch.equal(tr.v, tr2.v)
    for (k, a), (_, b) in zip(model.state_dict().items(), model2.state_dict().items()):
        assert torch.equal(a, b), k


@pytest.mark.parametrize("arch", ["MonoDepth2Model", "SupDepthModel"])
def test_training_loop_schedules_checkpoints_evaluates_and_resumes(tmp_path, arch):
    """engine.loops.do_train (the projects' do_train on the HIP path): LR rules, LOG_PERIOD records in metrics.json, a checkpoint per epoch,
    evaluation every epoch, and a run interrupted after epoch 0 that resumes to the same weights as the uninterrupted run, bit for bit."""
    import json
    import numpy as np
    from simpledepthestimation_amd.engine.loops import do_train
    from simpledepthestimation_amd.layers.fakeDDP import FakeDDP
    sup = arch == "SupDepthModel"
    mk = (lambda s: sup_batch(2, 64, 192, s)) if sup else (lambda s: mono_batch(2, 64, 192, s))
    loader = [mk(60 + i) for i in range(3)]
    rng = np.random.default_rng(1)
    test_loader = [{"img": loader[0]["img"], "metadata": [{"h_before_resize": 96, "w_before_resize": 288}] * 2,
                    "depth_orig": [np.where(rng.random((1, 96, 288)) < 0.3, rng.random((1, 96, 288)) * 60 + 2, 0).astype(np.float32) for _ in range(2)]}]

    def cfg_for(out, epochs):
        cfg = make_cfg(arch, 18)
        cfg.OUTPUT_DIR = str(out); cfg.LOG_PERIOD = 2; cfg.SOLVER.MAX_EPOCHS = epochs; cfg.SOLVER.LR_STEPS = (1,); cfg.TEST.EVAL_PERIOD = 1
        cfg.DATASETS.TEST.PREPROCESS = [{"NAME": "LoadImg"}, {"NAME": "Resize", "IMG_W": 192, "IMG_H": 64}, {"NAME": "ToTensor"}]   # the test loader's metadata
        cfg.TEST.GT_SCALE = False
        return cfg

    sd = OM.init_state_dict(18, with_pose=not sup, seed=33)
    full = FakeDDP(build(arch, 18, sd))
    rec = do_train(cfg_for(tmp_path / "full", 2), full, loader, test_loader)
    files = sorted(f for f in os.listdir(tmp_path / "full"))
    assert files == ["last_checkpoint", "metrics.json", "model_0000000.pth", "model_0000001.pth", "model_final.pth"]
    lines = [json.loads(l) for l in open(tmp_path / "full" / "metrics.json")]
    assert lines == rec and [r["iteration"] for r in rec] == [2, 3, 5, 6]            # LOG_PERIOD 2 of 3 batches + one evaluation record per epoch
    assert all(np.isfinite(r["total_loss"]) for r in rec if "total_loss" in r)
    assert "kitti evaluator/abs_rel" in rec[1] and "kitti evaluator (0-30m)/d1" in rec[3]
    cfg = cfg_for(tmp_path, 2)
    if sup:
        want = [(cfg.SOLVER.DEPTH_LR - cfg.SOLVER.DEPTH_END_LR) * (1 - g / 6) ** 0.9 + cfg.SOLVER.DEPTH_END_LR for g in (1, 4)]
        assert abs(rec[0]["lr"] - want[0]) < 1e-12 and abs(rec[2]["lr"] - want[1]) < 1e-12       # step g runs on f(g - 1)
    else:
        assert rec[0]["lr"] == cfg.SOLVER.DEPTH_LR and abs(rec[2]["lr"] - cfg.SOLVER.DEPTH_LR * cfg.SOLVER.GAMMA) < 1e-15   # milestone at epoch 1
    # a run that died after epoch 0: its directory holds what the uninterrupted run had written by then
    os.makedirs(tmp_path / "part")
    shutil.copy(tmp_path / "full" / "model_0000000.pth", tmp_path / "part" / "model_0000000.pth")
    with open(tmp_path / "part" / "last_checkpoint", "w") as f:
        f.write("model_0000000.pth")
    resumed = FakeDDP(build(arch, 18, OM.init_state_dict(18, with_pose=not sup, seed=34)))
    rec2 = do_train(cfg_for(tmp_path / "part", 2), resumed, loader, None, resume=True)
    assert [r["iteration"] for r in rec2] == [5] and rec2[0]["lr"] == rec[2]["lr"]
    for (k, a), (_, b) in zip(full.module.state_dict().items(), resumed.module.state_dict().items()):
        assert torch.equal(a, b), k


def test_graph_step_takes_the_collate_batch_dicts(tmp_path):
    """The batch dicts of data/datasets/kitti_v2.py:L196-221 through do_train with the hipGraph step: ctx_img / ctx_img_orig as lists of NUMPY
    arrays, `flip` as one python bool that changes from batch to batch (RandomFlip is in both projects' Base.yaml), `metadata` as a list of
    dicts.  Each flip value gets its own captured graph set; the run must equal the eager run bit for bit."""
    import numpy as np
    from simpledepthestimation_amd.engine.loops import do_train
    from simpledepthestimation_amd.layers.fakeDDP import FakeDDP

    def collated(seed, flip):
        b = mono_batch(2, 64, 192, seed)
        b["ctx_img"] = [t.numpy() for t in b["ctx_img"]]
        b["ctx_img_orig"] = [t.numpy() for t in b["ctx_img_orig"]]
        b["flip"] = flip
        b["metadata"] = [{"date": "2011_09_26", "drive": "0001", "img_id": f"{seed:010d}"}] * 2
        return b
    loader = [collated(70, False), collated(71, True), collated(72, True), collated(73, False)]
    finals = []
    for use_graph in (True, False):
        cfg = make_cfg("MonoDepth2Model", 18)
        cfg.OUTPUT_DIR = str(tmp_path / ("g" if use_graph else "e")); cfg.LOG_PERIOD = 2; cfg.SOLVER.MAX_EPOCHS = 2; cfg.TEST.EVAL_PERIOD = 0
        model = FakeDDP(build("MonoDepth2Model", 18, OM.init_state_dict(18, with_pose=True, seed=35)))
        rec = do_train(cfg, model, loader, None, use_graph=use_graph)
        assert [r["iteration"] for r in rec] == [2, 4, 6, 8] and all(np.isfinite(r["total_loss"]) for r in rec)
        finals.append({k: v.detach().clone() for k, v in model.module.state_dict().items()})
    for k in finals[0]:
        assert torch.equal(finals[0][k], finals[1][k]), k


def _amp_cfg(arch, enc):
    cfg = make_cfg(arch, enc, "fp16")
    cfg.SOLVER.AMP = True
    return cfg


def test_fp16_forward_tracks_fp32():
    """fp16 storage (BASELINE.json configs[4]'s precision): same weights, loss / depth close to the fp32 path (fp16 has 3 more mantissa bits
    than bf16, so the bounds are those of the bf16 test or tighter)."""
    sd = OM.init_state_dict(18, seed=3)
    batch = sup_batch(2, 64, 192, 4)
    m32 = build("SupDepthModel", 18, sd, "fp32").train()
    m16 = build("SupDepthModel", 18, sd, "fp16").train()
    o32 = m32(clone_batch(batch)); o16 = m16(clone_batch(batch))
    assert abs(o16["silog_loss"].item() - o32["silog_loss"].item()) < 1e-2 * o32["silog_loss"].item()
    assert rel(o16["depth_pred"][0], o32["depth_pred"][0]) < 1e-2


@pytest.mark.parametrize("arch,enc", [("SupDepthModel", 18), ("MonoDepth2Model", "packnet1A")])
def test_fp16_loss_scaling_steps_track_fp32_and_overflow_skips_the_step(arch, enc):
    """SOLVER.AMP (the reference's AMPTrainer / GradScaler, detectron2/engine/train_loop.py:L294-341) on the device: fp16 steps with dynamic loss
    scaling follow the fp32 trainer's losses; the scale backs off (x 0.5) while the scaled gradients overflow fp16 and such steps leave the
    parameters and the Adam moments untouched; an injected inf is caught, skips exactly that step and halves the scale; growth after
    `growth_interval` clean steps."""
    from simpledepthestimation_amd.engine.trainer import monodepth2_trainer, supervised_trainer
    sup = arch == "SupDepthModel"
    mk = supervised_trainer if sup else monodepth2_trainer
    sd = OM.init_state_dict(18, seed=3) if sup else OM.init_packnet_state_dict("A", seed=5)
    raw = sup_batch(2, 64, 192, 4) if sup else mono_batch(1, 64, 192, 21)
    batch = {k: ([x.to(dev) for x in v] if isinstance(v, list) else v.to(dev)) for k, v in raw.items()}
    t32 = mk(build(arch, enc, sd, "fp32").train(), make_cfg(arch, enc, "fp32"))
    t16 = mk(build(arch, enc, sd, "fp16").train(), _amp_cfg(arch, enc), growth_interval=3)
    assert t16.amp and t16.scale_state is not None and float(t16.scale_state[0]) == 65536.0
    with pytest.raises(ValueError):
        mk(build(arch, enc, sd, "bf16").train(), (lambda c: (setattr(c.SOLVER, "AMP", True), c)[1])(make_cfg(arch, enc, "bf16")))
    l32, l16, scales, moved = [], [], [], []
    for _ in range(8):
        p_before = t16.pflat.clone()
        l32.append(sum(float(v) for v in t32.step(clone_batch(batch)).values()))
        l16.append(sum(float(v) for v in t16.step(clone_batch(batch)).values()))
        scales.append(float(t16.scale_state[0]))
        moved.append(bool((t16.pflat != p_before).any()))
    assert all(x == x and abs(x) != float("inf") for x in l16), l16
    # every step either updated the parameters or backed the scale off; once the scale fits there are real updates
    prev = 65536.0
    for s, mv in zip(scales, moved):
        assert (s == prev * 0.5 and not mv) or (s >= prev and mv), (scales, moved)
        prev = s
    assert sum(moved) >= 3, (scales, moved)
    # the losses follow the fp32 run (skipped steps delay the fp16 trajectory: compare per number of applied updates)
    k = sum(moved)
    assert abs(l16[0] - l32[0]) < 2e-2 * abs(l32[0])
    applied16 = [l for l, mv in zip(l16[1:], moved[:-1]) if mv]                   # loss seen after each applied update
    for a, b in zip(applied16, l32[1:1 + len(applied16)]):
        assert abs(a - b) < 6e-2 * abs(b), (l16, l32, moved)
    # growth: three clean steps in a row double the scale
    clean_runs = [i for i in range(2, len(scales)) if moved[i] and moved[i - 1] and moved[i - 2]]
    if clean_runs:
        assert any(scales[i] == 2.0 * scales[i - 1] for i in clean_runs) or scales[-1] > min(scales), scales
    # injected overflow: one inf in the flat gradient -> the step is skipped (weights, moments) and the scale halves
    t16._fwd_bwd(clone_batch(batch))
    t16.gflat[7] = float("inf")
    p0, m0, v0, s0 = t16.pflat.clone(), t16.m.clone(), t16.v.clone(), float(t16.scale_state[0])
    t16._optimizer()
    assert torch.equal(t16.pflat, p0) and torch.equal(t16.m, m0) and torch.equal(t16.v, v0)
    assert float(t16.scale_state[0]) == 0.5 * s0 and float(t16.scale_state[1]) == 0.0
    # the scale travels with the optimizer state
    osd = t16.state_dict()
    assert osd["loss_scale"] == 0.5 * s0
    # Adam's `step` counts the APPLIED updates only (GradScaler.step skips optimizer.step() on overflow), not the calls
    assert t16.applied_steps() == sum(moved) and t16.t == len(moved) + 1
    assert all(float(e["step"]) == sum(moved) for e in osd["state"].values())


@pytest.mark.parametrize("enc,hits_expected", [(50, 12), (18, None)])
def test_residual_batchnorm_backward_rides_in_the_next_blocks_data_gradient(enc, hits_expected):
    """torchvision's Bottleneck (resnet_encoder.py:L88-99): out = relu(bn3(y) + identity) feeds conv1 of the next block and its skip path.  The residual form of
    sde_conv_dgrad_bnbwd lets that conv1's data gradient do bn3's whole backward reduce pass (12 of ResNet-50's 16 blocks): same losses, same gradients as the
    separate pass (the sums are formed from the same rounded gm values, in another order)."""
    from simpledepthestimation_amd.hip import nn as HN
    sd = OM.init_state_dict(enc, seed=5)
    batch = {k: v.to(dev) for k, v in sup_batch(2, 64, 192, 31).items()}
    res = []
    for fused in (False, True):
        kept, HN.RESBN_FUSED = HN.RESBN_FUSED, fused
        hits = HN.RESBN_HITS
        try:
            model = build("SupDepthModel", enc, sd, "bf16").train()
            out = model(clone_batch(batch))
            out["silog_loss"].backward()
            torch.cuda.synchronize()
        finally:
            HN.RESBN_FUSED = kept
        took = HN.RESBN_HITS - hits           # ResNet-50: 12 of its 16 blocks; ResNet-18: the blocks whose 3x3 conv1 runs on a kernel with the epilogue at this size
        assert took == ((hits_expected if hits_expected is not None else took) if fused else 0) and (not fused or took >= 1), took
        res.append((float(out["silog_loss"]), {n: p.grad.clone() for n, p in model.named_parameters() if p.grad is not None}))
    assert res[0][0] == res[1][0]
    # bf16 storage: the two paths round the same values in another order, and 50 BatchNorm'd layers amplify that towards the stem (its BatchNorm bias, a sum
    # with cancellation, moves most); the op-level test (test_gpu_nn.py) holds the kernel itself to 1e-5
    worst = max((rel(res[1][1][n], res[0][1][n]), n) for n in res[0][1])
    assert worst[0] < 6e-2, worst
    a = torch.cat([res[1][1][n].flatten().double() for n in res[0][1]]); b = torch.cat([res[0][1][n].flatten().double() for n in res[0][1]])
    assert float((a - b).norm() / b.norm()) < 3e-2          # (measured 1.1e-2: bf16 re-rounding noise of 16 blocks; the bf16-vs-fp32 test bounds both paths)
