"""Oracle (test infrastructure): SupDepthModel / MonoDepth2Model forward, torch-CPU fp32.

Restates meta_arch/Supervised.py:L36-49 and meta_arch/MonoDepth2.py:L55-128 on top of
oracle.nets / oracle.losses / oracle.geometry.  MonoDepth2 uses the intended
semantics for the translation (constant per sample).
"""
import math

import torch

from . import geometry as G
from . import losses as L
from . import nets as N

PIXEL_MEAN = [0.485, 0.456, 0.406]
PIXEL_STD = [0.229, 0.224, 0.225]


def init_state_dict(num_layers, with_pose=False, num_ctx=2, seed=0):
    """Random-init parameters with the reference's names and initialisers.

    Encoder: kaiming-normal(fan_out, relu) convs, BN gamma=1 beta=0 (torchvision ResNet);
    decoder: torch Conv2d default (kaiming-uniform a=sqrt(5), bias U(+-1/sqrt(fan_in)));
    PoseNet: xavier-uniform weights, zero bias (PoseNet.py:L42-48), GN gamma=1 beta=0.
    """
    g = torch.Generator().manual_seed(seed)
    sd = {}
    kind, reps = N.RESNET_SPECS[num_layers]
    exp = 1 if kind == "basic" else 4

    def conv_k(name, co, ci, k):
        std = math.sqrt(2.0 / (co * k * k))
        sd[name] = torch.randn(co, ci, k, k, generator=g) * std

    def bn(name, c):
        sd[name + ".weight"] = torch.ones(c)
        sd[name + ".bias"] = torch.zeros(c)
        sd[name + ".running_mean"] = torch.zeros(c)
        sd[name + ".running_var"] = torch.ones(c)
        sd[name + ".num_batches_tracked"] = torch.zeros((), dtype=torch.long)

    p = "depth_net.encoder.encoder."
    conv_k(p + "conv1.weight", 64, 3, 7)
    bn(p + "bn1", 64)
    inpl = 64
    for li, n in enumerate(reps):
        planes = 64 * 2 ** li
        for b in range(n):
            q = f"{p}layer{li + 1}.{b}"
            stride = 2 if (b == 0 and li > 0) else 1
            if kind == "basic":
                conv_k(q + ".conv1.weight", planes, inpl, 3); bn(q + ".bn1", planes)
                conv_k(q + ".conv2.weight", planes, planes, 3); bn(q + ".bn2", planes)
            else:
                conv_k(q + ".conv1.weight", planes, inpl, 1); bn(q + ".bn1", planes)
                conv_k(q + ".conv2.weight", planes, planes, 3); bn(q + ".bn2", planes)
                conv_k(q + ".conv3.weight", planes * 4, planes, 1); bn(q + ".bn3", planes * 4)
            if stride != 1 or inpl != planes * exp:
                conv_k(q + ".downsample.0.weight", planes * exp, inpl, 1); bn(q + ".downsample.1", planes * exp)
            inpl = planes * exp
    # torchvision's unused classifier head (kept for state-dict compatibility)
    bound = 1.0 / math.sqrt(512 * exp)
    sd[p + "fc.weight"] = (torch.rand(1000, 512 * exp, generator=g) * 2 - 1) * bound
    sd[p + "fc.bias"] = (torch.rand(1000, generator=g) * 2 - 1) * bound

    def conv_default(wname, bname, co, ci, k):
        fan_in = ci * k * k
        bound = 1.0 / math.sqrt(fan_in)
        sd[wname] = (torch.rand(co, ci, k, k, generator=g) * 2 - 1) * bound
        sd[bname] = (torch.rand(co, generator=g) * 2 - 1) * bound

    p = "depth_net.decoder.decoder."
    for k, (name, ci, co) in enumerate(N.decoder_layout(num_layers)):
        mid = ".conv.conv." if name[0] == "upconv" else ".conv."
        conv_default(f"{p}{k}{mid}weight", f"{p}{k}{mid}bias", co, ci, 3)

    if with_pose:
        p = "pose_net."
        cin = 3 * (1 + num_ctx)
        for i, (co, k) in enumerate(zip(N.POSE_CH, N.POSE_K)):
            fan_in, fan_out = cin * k * k, co * k * k
            a = math.sqrt(6.0 / (fan_in + fan_out))
            sd[f"{p}conv{i + 1}.0.weight"] = (torch.rand(co, cin, k, k, generator=g) * 2 - 1) * a
            sd[f"{p}conv{i + 1}.0.bias"] = torch.zeros(co)
            sd[f"{p}conv{i + 1}.1.weight"] = torch.ones(co)
            sd[f"{p}conv{i + 1}.1.bias"] = torch.zeros(co)
            cin = co
        a = math.sqrt(6.0 / (256 + 6 * num_ctx))
        sd[p + "pose_pred.weight"] = (torch.rand(6 * num_ctx, 256, 1, 1, generator=g) * 2 - 1) * a
        sd[p + "pose_pred.bias"] = torch.zeros(6 * num_ctx)
    sd["pixel_mean"] = torch.tensor(PIXEL_MEAN).view(1, 3, 1, 1)
    sd["pixel_std"] = torch.tensor(PIXEL_STD).view(1, 3, 1, 1)
    return sd


def init_packnet_state_dict(version="A", with_pose=True, num_ctx=2, seed=0):
    """PackNet01 (+ PoseNet) parameters with the reference's names and initialisers: xavier-uniform Conv2d / Conv3d weights, zero
    biases (PackNet01.py:L110-116), GroupNorm gamma=1 beta=0."""
    g = torch.Generator().manual_seed(seed)
    sd = {}
    p = "depth_net."

    def xavier(name, shape):
        rf = 1
        for s_ in shape[2:]:
            rf *= s_
        a = math.sqrt(6.0 / (shape[1] * rf + shape[0] * rf))
        sd[name + ".weight"] = (torch.rand(*shape, generator=g) * 2 - 1) * a
        sd[name + ".bias"] = torch.zeros(shape[0])

    def gn(name, c):
        sd[name + ".weight"] = torch.ones(c); sd[name + ".bias"] = torch.zeros(c)

    def conv2d(name, ci, co, k):
        xavier(name + ".conv_base", (co, ci, k, k)); gn(name + ".normalize", co)

    for name, kind, a in N.packnet_layout(version):
        if kind == "conv2d":
            conv2d(p + name, a[0], a[1], a[2])
        elif kind == "pack":
            conv2d(p + name + ".conv", a[0] * 4 * 8, a[0], a[1]); xavier(p + name + ".conv3d", (8, 1, 3, 3, 3))
        elif kind == "unpack":
            conv2d(p + name + ".conv", a[0], a[1] * 4 // 8, a[2]); xavier(p + name + ".conv3d", (8, 1, 3, 3, 3))
        elif kind == "res":
            conv2d(p + name + ".conv1", a[0], a[1], 3); conv2d(p + name + ".conv2", a[1], a[1], 3)
            xavier(p + name + ".conv3", (a[1], a[0], 1, 1)); gn(p + name + ".normalize", a[1])
        else:
            xavier(p + name + ".conv1", (1, a[0], 3, 3))
    if with_pose:
        pose = init_state_dict(18, with_pose=True, num_ctx=num_ctx, seed=seed + 1)
        sd.update({k: v for k, v in pose.items() if k.startswith("pose_net.")})
    sd["pixel_mean"] = torch.tensor(PIXEL_MEAN).view(1, 3, 1, 1)
    sd["pixel_std"] = torch.tensor(PIXEL_STD).view(1, 3, 1, 1)
    return sd


def normalise(sd, img):
    return (img - sd["pixel_mean"]) / sd["pixel_std"]


def supervised_forward(sd, batch, num_layers, max_depth=80.0, variance_focus=0.85, training=True,
                       update_running=False):
    """Supervised.py:L36-49."""
    x = normalise(sd, batch["img"])
    depths, _ = N.depth_resnet(sd, x, num_layers, max_depth, bool(batch.get("flip", False)), training, update_running)
    if not training:
        return {"depth_pred": depths[0]}
    losses = []
    for d in depths:
        gt = G.resize_img(batch["depth"], d.shape[-2:], mode="nearest")
        losses.append(L.silog(d, gt, variance_focus))
    return {"silog_loss": sum(losses) / len(losses), "depth_pred": depths}


def monodepth2_losses(depths, poses, image, contexts, intrinsics, ssim_w=0.85, C1=1e-4, C2=9e-4, automask=True,
                      smooth_w=1e-3, reduce="min", var_w=0.0, clip=0.0):
    """MonoDepth2.py:L67-124 given the network outputs.

    depths: 4 x [B,1,h,w]; poses: list of [B,4,4]; image/contexts: original frames.
    """
    ns = len(depths)
    out = {}
    photo = []
    smooth = 0.0
    var = 0.0
    for i, d in enumerate(depths):
        hw = d.shape[-2:]
        scale_w = 1.0 / 2 ** (ns - i - 1)
        img_i = G.resize_img(image, hw)
        K_i = G.scale_intrinsics(intrinsics, hw[1] / image.shape[-1], hw[0] / image.shape[-2])
        maps = []
        for ctx, pose in zip(contexts, poses):
            ctx_i = G.resize_img(ctx, hw)
            maps.append(L.rgb_consistency(img_i, ctx_i, d, K_i, pose[:, :3, :3], pose[:, :3, 3],
                                          ssim_w=ssim_w, C1=C1, C2=C2, clip=clip))
            if automask:
                maps.append(L.rgb_consistency(img_i, ctx_i, d, K_i, None, None, ssim_w=ssim_w, C1=C1, C2=C2, clip=clip))
        if reduce == "min":
            photo.append(L.min_reprojection(maps))
        else:
            photo.append(sum(m.mean() for m in maps) / len(maps))
        if smooth_w > 0.0:
            smooth = smooth + L.smoothness(d, img_i) * scale_w * smooth_w / ns
        if var_w > 0.0:
            var = var + L.variance(d) * scale_w * var_w / ns
    out["rec_loss"] = sum(photo) / ns
    if smooth_w > 0.0:
        out["smooth_loss"] = smooth
    if var_w > 0.0:
        out["var_loss"] = var
    return out


def monodepth2_forward(sd, batch, num_layers, max_depth=80.0, training=True, update_running=False, upsample_depth=False, **loss_kw):
    """MonoDepth2.py:L55-128.  num_layers: 18 / 34 / 50 (DepthResNet) or "packnet1A" / "packnet1B" (PackNet01)."""
    x = normalise(sd, batch["img"])
    if isinstance(num_layers, str) and num_layers.startswith("packnet"):
        depths = N.packnet01(sd, x, num_layers[-1], max_depth, bool(batch.get("flip", False)))
    else:
        depths, _ = N.depth_resnet(sd, x, num_layers, max_depth, bool(batch.get("flip", False)), training, update_running,
                                   upsample_depth=upsample_depth)
    if not training:
        return {"depth_pred": depths[0]}
    pin = torch.cat([batch["img"]] + list(batch["ctx_img"]), 1)
    vec = N.pose_net(sd, pin, len(batch["ctx_img"]))
    poses = [G.pose_vec2mat(vec[:, j]) for j in range(vec.shape[1])]
    out = monodepth2_losses(depths, poses, batch["img_orig"], batch["ctx_img_orig"], batch["intrinsics"], **loss_kw)
    out["depth_pred"] = depths
    out["pose_vec"] = vec
    return out
