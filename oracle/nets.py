"""Oracle (test infrastructure): DepthResNet / PoseNet restated functionally on torch-CPU.

Every function takes a flat ``{state_dict_key: tensor}`` mapping that uses the
reference's parameter names (checkpoint compatibility), so the
product model's ``state_dict()`` can be fed in unchanged.

Reference call sites restated here:
  * layers/resnet_encoder.py:L88-99   (5-feature encoder forward)
  * layers/depth_decoder.py:L9-110    (upconv / skip-fusion / dispconv + softplus)
  * modeling/depth_net/DepthResNet.py:L45-70
  * modeling/pose_net/PoseNet.py:L13-65

PARITY UNPINNED for the ResNet wiring itself: the reference delegates it to
torchvision 0.9.0 (README.md:L26, not vendored, not installed here).  It is restated
from the published architecture (ResNet v1.5: stride on the 3x3 of a Bottleneck);
its arithmetic is torch's own CPU conv2d / batch_norm.  The goldens for the encoder
come from running the reference's resnet_encoder.py against this same restatement
(oracle/ref_harness.py), so they pin the reference-side glue, not torchvision.
"""
import torch
import torch.nn.functional as F

RESNET_SPECS = {
    18: ("basic", [2, 2, 2, 2]),
    34: ("basic", [3, 4, 6, 3]),
    50: ("bottleneck", [3, 4, 6, 3]),
}
NUM_CH_DEC = [16, 32, 64, 128, 256]


def num_ch_enc(num_layers):
    base = [64, 64, 128, 256, 512]
    return base if num_layers <= 34 else [base[0]] + [c * 4 for c in base[1:]]


def _bn(sd, p, x, training, momentum=0.1, eps=1e-5, update_running=False):
    rm, rv = sd[p + ".running_mean"], sd[p + ".running_var"]
    if training and not update_running:
        rm, rv = rm.clone(), rv.clone()
    return F.batch_norm(x, rm, rv, sd[p + ".weight"], sd[p + ".bias"], training, momentum, eps)


def _basic_block(sd, p, x, stride, training, upd):
    idt = x
    out = F.conv2d(x, sd[p + ".conv1.weight"], None, stride, 1)
    out = F.relu(_bn(sd, p + ".bn1", out, training, update_running=upd))
    out = F.conv2d(out, sd[p + ".conv2.weight"], None, 1, 1)
    out = _bn(sd, p + ".bn2", out, training, update_running=upd)
    if (p + ".downsample.0.weight") in sd:
        idt = F.conv2d(x, sd[p + ".downsample.0.weight"], None, stride, 0)
        idt = _bn(sd, p + ".downsample.1", idt, training, update_running=upd)
    return F.relu(out + idt)


def _bottleneck(sd, p, x, stride, training, upd):
    idt = x
    out = F.relu(_bn(sd, p + ".bn1", F.conv2d(x, sd[p + ".conv1.weight"]), training, update_running=upd))
    out = F.conv2d(out, sd[p + ".conv2.weight"], None, stride, 1)
    out = F.relu(_bn(sd, p + ".bn2", out, training, update_running=upd))
    out = _bn(sd, p + ".bn3", F.conv2d(out, sd[p + ".conv3.weight"]), training, update_running=upd)
    if (p + ".downsample.0.weight") in sd:
        idt = F.conv2d(x, sd[p + ".downsample.0.weight"], None, stride, 0)
        idt = _bn(sd, p + ".downsample.1", idt, training, update_running=upd)
    return F.relu(out + idt)


def resnet_encoder(sd, prefix, x, num_layers, training=True, update_running=False):
    """resnet_encoder.py:L88-99 -> [relu(bn1(conv1)), layer1(maxpool), layer2, layer3, layer4]."""
    kind, reps = RESNET_SPECS[num_layers]
    blk = _basic_block if kind == "basic" else _bottleneck
    p = prefix
    f = F.conv2d(x, sd[p + "conv1.weight"], None, 2, 3)
    f = F.relu(_bn(sd, p + "bn1", f, training, update_running=update_running))
    feats = [f]
    y = F.max_pool2d(f, 3, 2, 1)
    for li, n in enumerate(reps):
        for b in range(n):
            stride = 2 if (b == 0 and li > 0) else 1
            y = blk(sd, f"{p}layer{li + 1}.{b}", y, stride, training, update_running)
        feats.append(y)
    return feats


def decoder_layout(num_layers):
    """ModuleList order of depth_decoder.py:L76-92: upconv(4,0),(4,1),...,(0,1), dispconv 0..3."""
    enc = num_ch_enc(num_layers)
    layout = []
    for i in range(4, -1, -1):
        cin = enc[-1] if i == 4 else NUM_CH_DEC[i + 1]
        layout.append((("upconv", i, 0), cin, NUM_CH_DEC[i]))
        cin = NUM_CH_DEC[i] + (enc[i - 1] if i > 0 else 0)
        layout.append((("upconv", i, 1), cin, NUM_CH_DEC[i]))
    for s in range(4):
        layout.append((("dispconv", s), NUM_CH_DEC[s], 1))
    return layout


def _conv3x3_refl(x, w, b):
    return F.conv2d(F.pad(x, (1, 1, 1, 1), mode="reflect"), w, b)


def depth_decoder(sd, prefix, feats, num_layers):
    """depth_decoder.py:L95-110 -> {scale: softplus(dispconv)} for scales 0..3."""
    idx = {name: k for k, (name, _, _) in enumerate(decoder_layout(num_layers))}

    def block(name, x):
        k = idx[name]
        if name[0] == "upconv":
            return F.elu(_conv3x3_refl(x, sd[f"{prefix}{k}.conv.conv.weight"], sd[f"{prefix}{k}.conv.conv.bias"]))
        return _conv3x3_refl(x, sd[f"{prefix}{k}.conv.weight"], sd[f"{prefix}{k}.conv.bias"])

    out = {}
    x = feats[-1]
    for i in range(4, -1, -1):
        x = block(("upconv", i, 0), x)
        x = F.interpolate(x, scale_factor=2, mode="nearest")
        if i > 0:
            x = torch.cat([x, feats[i - 1]], 1)
        x = block(("upconv", i, 1), x)
        if i < 4:
            out[i] = F.softplus(block(("dispconv", i), x))
    return out


def disp_to_depth(disp, min_depth, max_depth):
    """depth_decoder.py:L9-18."""
    min_disp, max_disp = 1 / max_depth, 1 / min_depth
    scaled = min_disp + (max_disp - min_disp) * disp
    return scaled, 1 / scaled


def depth_resnet(sd, x, num_layers, max_depth=80.0, flip=False, training=True, update_running=False,
                 prefix="depth_net.", upsample_depth=False):
    """DepthResNet.py:L45-70 -> list of 4 depth maps (index 0 = full resolution)."""
    if flip:
        x = torch.flip(x, [3])
    feats = resnet_encoder(sd, prefix + "encoder.encoder.", x, num_layers, training, update_running)
    disps = depth_decoder(sd, prefix + "decoder.decoder.", feats, num_layers)
    depths = [disp_to_depth(disps[i], 0.1, max_depth)[1] for i in range(4)]
    if flip:
        depths = [torch.flip(d, [3]) for d in depths]
    if upsample_depth:       # DepthResNet.py:L62-63
        depths = [F.interpolate(d, size=x.shape[-2:], mode="nearest") for d in depths]
    return depths, feats


POSE_CH = [16, 32, 64, 128, 256, 256, 256]
POSE_K = [7, 5, 3, 3, 3, 3, 3]


def pose_net(sd, x, num_ctx=2, prefix="pose_net."):
    """PoseNet.py:L50-65 -> pose vectors [B,num_ctx,6] (already x0.01)."""
    for i in range(7):
        k = POSE_K[i]
        x = F.conv2d(x, sd[f"{prefix}conv{i + 1}.0.weight"], sd[f"{prefix}conv{i + 1}.0.bias"], 2, (k - 1) // 2)
        x = F.group_norm(x, 16, sd[f"{prefix}conv{i + 1}.1.weight"], sd[f"{prefix}conv{i + 1}.1.bias"], 1e-5)
        x = F.relu(x)
    p = F.conv2d(x, sd[prefix + "pose_pred.weight"], sd[prefix + "pose_pred.bias"])
    p = p.mean(3).mean(2)
    return 0.01 * p.view(p.size(0), num_ctx, 6)


# ------------------------------------------------------------------------------------------------------------------
# PackNet01 (reference: modeling/depth_net/PackNet01.py:L18-209 + layers/layers01.py:L11-298), version "A" / "B"
# ------------------------------------------------------------------------------------------------------------------
def packnet_layout(version="A"):
    """(module path, kind, args) for every parameterised layer, in the reference's construction order.
    kinds: conv2d (Conv2D: conv_base + GroupNorm), res (ResidualConv), pack / unpack (… + conv3d), inv (InvDepth)."""
    ni, no = 64, 1
    n1, n2, n3, n4, n5 = 64, 64, 128, 256, 512
    if version == "A":
        n1o, n1i = n1, n1 + ni + no
        n2o, n2i = n2, n2 + n1 + no
        n3o, n3i = n3, n3 + n2 + no
        n4o, n4i = n4, n4 + n3
        n5o, n5i = n5, n5 + n4
    else:
        n1o, n1i = n1, n1 + no
        n2o, n2i = n2, n2 + no
        n3o, n3i = n3 // 2, n3 // 2 + no
        n4o, n4i = n4 // 2, n4 // 2
        n5o, n5i = n5 // 2, n5 // 2
    L = [("pre_calc", "conv2d", (3, ni, 5))]
    for name, c, k in [("pack1", n1, 5), ("pack2", n2, 3), ("pack3", n3, 3), ("pack4", n4, 3), ("pack5", n5, 3)]:
        L.append((name, "pack", (c, k)))
    L.append(("conv1", "conv2d", (ni, n1, 7)))
    for name, ci, co, nb in [("conv2", n1, n2, 2), ("conv3", n2, n3, 2), ("conv4", n3, n4, 3), ("conv5", n4, n5, 3)]:
        for b in range(nb):
            L.append((f"{name}.{b}", "res", (ci if b == 0 else co, co)))
    for name, ci, co in [("unpack5", n5, n5o), ("unpack4", n5, n4o), ("unpack3", n4, n3o), ("unpack2", n3, n2o), ("unpack1", n2, n1o)]:
        L.append((name, "unpack", (ci, co, 3)))
    for name, ci, co in [("iconv5", n5i, n5), ("iconv4", n4i, n4), ("iconv3", n3i, n3), ("iconv2", n2i, n2), ("iconv1", n1i, n1)]:
        L.append((name, "conv2d", (ci, co, 3)))
    for name, ci in [("disp4_layer", n4), ("disp3_layer", n3), ("disp2_layer", n2), ("disp1_layer", n1)]:
        L.append((name, "inv", (ci,)))
    return L


def _pn_conv2d(sd, p, x):
    """layers01.Conv2D: ConstantPad2d(k//2) + Conv2d + GroupNorm(16) + ELU."""
    w = sd[p + ".conv_base.weight"]
    x = F.conv2d(x, w, sd[p + ".conv_base.bias"], 1, w.shape[-1] // 2)
    return F.elu(F.group_norm(x, 16, sd[p + ".normalize.weight"], sd[p + ".normalize.bias"], 1e-5))


def _pn_res(sd, p, x):
    out = _pn_conv2d(sd, p + ".conv2", _pn_conv2d(sd, p + ".conv1", x))
    sc = F.conv2d(x, sd[p + ".conv3.weight"], sd[p + ".conv3.bias"])
    return F.elu(F.group_norm(out + sc, 16, sd[p + ".normalize.weight"], sd[p + ".normalize.bias"], 1e-5))


def packing(x, r=2):
    """layers01.py:L138-160."""
    b, c, h, w = x.shape
    x = x.contiguous().view(b, c, h // r, r, w // r, r)
    return x.permute(0, 1, 3, 5, 2, 4).contiguous().view(b, c * r * r, h // r, w // r)


def _pn_conv3d(sd, p, x):
    y = F.conv3d(x.unsqueeze(1), sd[p + ".conv3d.weight"], sd[p + ".conv3d.bias"], padding=1)
    b, c, d, h, w = y.shape
    return y.view(b, c * d, h, w)


def _pn_pack(sd, p, x):
    return _pn_conv2d(sd, p + ".conv", _pn_conv3d(sd, p, packing(x)))


def _pn_unpack(sd, p, x):
    return F.pixel_shuffle(_pn_conv3d(sd, p, _pn_conv2d(sd, p + ".conv", x)), 2)


def _pn_inv(sd, p, x):
    return torch.sigmoid(F.conv2d(x, sd[p + ".conv1.weight"], sd[p + ".conv1.bias"], 1, 1)) / 0.5


def packnet01(sd, x, version="A", max_depth=80.0, flip=False, prefix="depth_net."):
    """PackNet01.forward (PackNet01.py:L118-209) -> 4 metric depth maps (index 0 = full resolution)."""
    q = lambda n: prefix + n   # noqa: E731
    if flip:
        x = torch.flip(x, [3])
    x = _pn_conv2d(sd, q("pre_calc"), x)
    x1 = _pn_conv2d(sd, q("conv1"), x)
    x1p = _pn_pack(sd, q("pack1"), x1)

    def block(name, n, t):
        for b in range(n):
            t = _pn_res(sd, q(f"{name}.{b}"), t)
        return t
    x2p = _pn_pack(sd, q("pack2"), block("conv2", 2, x1p))
    x3p = _pn_pack(sd, q("pack3"), block("conv3", 2, x2p))
    x4p = _pn_pack(sd, q("pack4"), block("conv4", 3, x3p))
    x5p = _pn_pack(sd, q("pack5"), block("conv5", 3, x4p))
    skip1, skip2, skip3, skip4, skip5 = x, x1p, x2p, x3p, x4p

    def join(u, s, ud=None):
        parts = [u, s] if version == "A" else [u + s]
        if ud is not None:
            parts.append(ud)
        return torch.cat(parts, 1) if len(parts) > 1 else parts[0]
    up = lambda t: F.interpolate(t, scale_factor=2, mode="nearest")   # noqa: E731
    iconv5 = _pn_conv2d(sd, q("iconv5"), join(_pn_unpack(sd, q("unpack5"), x5p), skip5))
    iconv4 = _pn_conv2d(sd, q("iconv4"), join(_pn_unpack(sd, q("unpack4"), iconv5), skip4))
    disp4 = _pn_inv(sd, q("disp4_layer"), iconv4)
    iconv3 = _pn_conv2d(sd, q("iconv3"), join(_pn_unpack(sd, q("unpack3"), iconv4), skip3, up(disp4)))
    disp3 = _pn_inv(sd, q("disp3_layer"), iconv3)
    iconv2 = _pn_conv2d(sd, q("iconv2"), join(_pn_unpack(sd, q("unpack2"), iconv3), skip2, up(disp3)))
    disp2 = _pn_inv(sd, q("disp2_layer"), iconv2)
    iconv1 = _pn_conv2d(sd, q("iconv1"), join(_pn_unpack(sd, q("unpack1"), iconv2), skip1, up(disp2)))
    disp1 = _pn_inv(sd, q("disp1_layer"), iconv1)
    depths = [disp_to_depth(d, 0.1, max_depth)[1] for d in (disp1, disp2, disp3, disp4)]
    if flip:
        depths = [torch.flip(d, [3]) for d in depths]
    return depths
