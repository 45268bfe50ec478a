"""Generate tests/golden/*.npz by running the REFERENCE's own code (build container only).

    python -m oracle.gen_golden

Inputs are seeded (torch CPU generator, deterministic across hosts) and small inputs are stored
next to the expected outputs; network weights come from oracle.models.init_state_dict(seed) and
are loaded into the reference modules with ``load_state_dict(strict=True)`` (which also pins the
state-dict key names).  Nothing from the reference's source text is written out -- only arrays.
"""
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from oracle import ref_harness, models as OM  # noqa: E402

OUT = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests", "golden")


from simpledepthestimation_amd.data.synthetic import kitti_K, mono_batch, smooth_images, sup_batch  # noqa: E402,F401  (one definition of the inputs)


POSE_VECS = torch.tensor([[0.05, -0.01, 0.3, 0.002, -0.004, 0.001],
                          [-0.03, 0.02, -0.25, -0.001, 0.003, 0.002]])


def np_(t):
    return t.detach().cpu().numpy().copy()


def gen_geometry(ref):
    cam, pu = ref.camera, ref.pose_utils
    ssim = ref.ssim_loss.SSIM(1e-4, 9e-4)
    res = {}
    for tag, (B, H, W, seed) in {"s48": (2, 48, 160, 1234), "s24": (2, 24, 80, 77), "full": (2, 192, 640, 1234)}.items():
        torch.manual_seed(seed)
        A = torch.rand(B, 3, H, W); Bf = torch.rand(B, 3, H, W); D = torch.rand(B, 1, H, W) * 79 + 1
        K = kitti_K(B, H, W)
        P = pu.pose_vec2mat(POSE_VECS)
        R = P[:, :3, :3]; t = P[:, :3, [3], None].expand(-1, -1, H, W)
        sampled, Z, grid, valid = cam.view_synthesis(Bf, D, K, R, t)
        ix = ((grid[..., 0] + 1) / 2) * (W - 1); iy = ((grid[..., 1] + 1) / 2) * (H - 1)
        fx, fy = ix.floor().to(torch.int32), iy.floor().to(torch.int32)
        s_map = ssim(sampled, A)
        l1 = (sampled - A).abs().mean(1, True)
        photo = s_map.mean(1, True) * 0.85 + l1 * 0.15
        l1i = (Bf - A).abs().mean(1, True)
        photo_id = ssim(Bf, A).mean(1, True) * 0.85 + l1i * 0.15
        mn = torch.cat([photo, photo_id], 1).min(1, True)[0].mean()
        smooth = ref.smoothness_loss.smoothness_loss(D, A)
        gt = torch.where(torch.rand(B, 1, H, W) < 0.3, torch.rand(B, 1, H, W) * 79 + 1, torch.zeros(1))
        sil = ref.losses.silog_loss(0.85)(D, gt)
        var = ref.losses.variance_loss(D)
        d = {"P": P, "sampled_mean": sampled.mean(), "Z_mean": Z.mean(), "valid_frac": valid.float().mean(),
             "sum_fx": fx.long().sum(), "sum_fy": fy.long().sum(), "ssim_mean": s_map.mean(),
             "photo_mean": photo.mean(), "min_mean": mn, "smooth": smooth, "silog": sil, "var": var,
             "grid_probe": grid[0, H // 2 + 4, W // 2 - 20]}
        if tag != "full":
            d.update({"A": A, "Bf": Bf, "D": D, "K": K, "gt": gt, "sampled": sampled, "Z": Z, "grid": grid,
                      "valid": valid, "fx": fx, "fy": fy, "ssim": s_map, "photo": photo, "photo_id": photo_id})
        else:
            d.update({"fx_sub": fx[:, ::8, ::8].contiguous(), "fy_sub": fy[:, ::8, ::8].contiguous(),
                      "sampled_sub": sampled[:, :, ::8, ::8].contiguous()})
        for k, v in d.items():
            res[f"{tag}.{k}"] = np_(v) if torch.is_tensor(v) else np.asarray(v)
    # a hard warp case: large rotation/translation so many samples leave the image / go behind the camera
    torch.manual_seed(5)
    B, H, W = 2, 24, 80
    Bf = torch.rand(B, 3, H, W); D = torch.rand(B, 1, H, W) * 5 + 0.05
    K = kitti_K(B, H, W)
    vec = torch.tensor([[1.5, -0.6, -3.0, 0.2, -0.3, 0.1], [-2.0, 0.4, 0.8, -0.1, 0.25, -0.4]])
    P = pu.pose_vec2mat(vec)
    sampled, Z, grid, valid = cam.view_synthesis(Bf, D, K, P[:, :3, :3], P[:, :3, [3], None].expand(-1, -1, H, W))
    ix = ((grid[..., 0] + 1) / 2) * (W - 1); iy = ((grid[..., 1] + 1) / 2) * (H - 1)
    for k, v in {"Bf": Bf, "D": D, "K": K, "vec": vec, "P": P, "sampled": sampled, "Z": Z, "valid": valid,
                 "fx": ix.floor().to(torch.int32), "fy": iy.floor().to(torch.int32), "grid": grid}.items():
        res[f"hard.{k}"] = np_(v)
    # resize / intrinsics / misc
    torch.manual_seed(9)
    img = torch.rand(1, 3, 48, 160); dep = torch.rand(1, 1, 48, 160) * 80
    for (h, w) in [(24, 80), (12, 40), (6, 20)]:
        res[f"resize.bil_{h}"] = np_(cam.resize_img(img, (h, w)))
        res[f"resize.nn_{h}"] = np_(cam.resize_img(dep, (h, w), mode="nearest"))
    res["resize.img"] = np_(img); res["resize.dep"] = np_(dep)
    Ks = cam.scale_intrinsics(kitti_K(2, 192, 640).clone(), 0.25, 0.25)
    res["misc.K_scaled"] = np_(Ks); res["misc.K_inv"] = np_(cam.inv_intrinsics(Ks))
    res["misc.vec"] = np_(torch.cat([POSE_VECS, vec], 0)); res["misc.P"] = np_(pu.pose_vec2mat(torch.cat([POSE_VECS, vec], 0)))
    sd, dd = ref.depth_decoder.disp_to_depth(torch.tensor(0.5), 0.1, 80)
    res["misc.disp_to_depth"] = np.array([float(sd), float(dd)], dtype=np.float64)
    np.savez_compressed(os.path.join(OUT, "geometry.npz"), **res)
    print("geometry.npz", len(res), "arrays")


def load_ref_weights(module, sd, prefix=""):
    own = {k[len(prefix):]: v.clone() for k, v in sd.items() if k.startswith(prefix)}
    missing, unexpected = module.load_state_dict(own, strict=True), None
    return missing


def grad_norms(model, names):
    out = {}
    named = dict(model.named_parameters())
    for n in names:
        out[n] = float(named[n].grad.norm())
    return out


PROBE_PARAMS = ["depth_net.encoder.encoder.conv1.weight",
                "depth_net.encoder.encoder.layer2.0.conv1.weight",
                "depth_net.encoder.encoder.layer4.1.bn2.weight",
                "depth_net.decoder.decoder.0.conv.conv.weight",
                "depth_net.decoder.decoder.9.conv.conv.bias",
                "depth_net.decoder.decoder.10.conv.weight"]
PROBE_PARAMS_50 = ["depth_net.encoder.encoder.conv1.weight",
                   "depth_net.encoder.encoder.layer2.0.conv2.weight",
                   "depth_net.encoder.encoder.layer4.2.bn3.weight",
                   "depth_net.decoder.decoder.1.conv.conv.weight",
                   "depth_net.decoder.decoder.13.conv.bias"]


def gen_models(ref):
    res = {}
    # --- Supervised (config 1 of BASELINE.json: R18, bs=2, 64x192, CPU) and an R50 case ---
    for tag, enc, B, H, W, probes in [("sup18", 18, 2, 64, 192, PROBE_PARAMS), ("sup50", 50, 1, 64, 192, PROBE_PARAMS_50)]:
        sd = OM.init_state_dict(enc, seed=100 + enc)
        torch.manual_seed(0)
        model = ref.Supervised.SupDepthModel(ref_harness.make_cfg("SupDepthModel", str(enc)))
        load_ref_weights(model, sd)
        model.train()
        batch = sup_batch(B, H, W, 3)
        out = model({k: v.clone() for k, v in batch.items()})
        out["silog_loss"].backward()
        res[f"{tag}.silog_loss"] = np.float64(out["silog_loss"].item())
        for i, d in enumerate(out["depth_pred"]):
            res[f"{tag}.depth{i}"] = np_(d)
        for n, v in grad_norms(model, probes).items():
            res[f"{tag}.gnorm.{n}"] = np.float64(v)
        res[f"{tag}.bn1_running_mean"] = np_(model.depth_net.encoder.encoder.bn1.running_mean)
        res[f"{tag}.bn1_running_var"] = np_(model.depth_net.encoder.encoder.bn1.running_var)
        model.eval()
        with torch.no_grad():
            res[f"{tag}.eval_depth"] = np_(model({k: v.clone() for k, v in batch.items()})["depth_pred"])
        # flip branch (DepthResNet.py:L52-60)
        model.train()
        fb = {k: v.clone() for k, v in batch.items()}; fb["flip"] = True
        with torch.no_grad():
            res[f"{tag}.flip_depth0"] = np_(model(fb)["depth_pred"][0])
        # encoder features + decoder alone
        if enc == 18:
            model.train()
            with torch.no_grad():
                x = (batch["img"] - model.pixel_mean) / model.pixel_std
                feats = model.depth_net.encoder(x)
                for i, f in enumerate(feats):
                    res[f"{tag}.feat{i}_mean"] = np.float64(f.mean().item())
                    res[f"{tag}.feat{i}_absmean"] = np.float64(f.abs().mean().item())
                res[f"{tag}.feat4"] = np_(feats[4])
                g = torch.Generator().manual_seed(11)
                rf = [torch.randn(f.shape, generator=g) for f in feats]
                disp = model.depth_net.decoder(rf)
                for i in range(4):
                    res[f"{tag}.dec_disp{i}"] = np_(disp[("disp", i)])
    # --- MonoDepth2 (R18), 64x192 with everything stored, 192x640 scalars only ---
    for tag, B, H, W in [("mono18", 2, 64, 192), ("mono18_full", 2, 192, 640)]:
        sd = OM.init_state_dict(18, with_pose=True, seed=7)
        model = ref.MonoDepth2Fixed(ref_harness.make_cfg("MonoDepth2Model", "18"))
        load_ref_weights(model, sd)
        model.train()
        batch = mono_batch(B, H, W, 21)
        out = model({k: ([x.clone() for x in v] if isinstance(v, list) else v.clone()) for k, v in batch.items()})
        loss = out["rec_loss"] + out["smooth_loss"]
        loss.backward()
        res[f"{tag}.rec_loss"] = np.float64(out["rec_loss"].item())
        res[f"{tag}.smooth_loss"] = np.float64(out["smooth_loss"].item())
        for n, v in grad_norms(model, PROBE_PARAMS + ["pose_net.conv1.0.weight", "pose_net.conv7.1.weight",
                                                     "pose_net.pose_pred.weight"]).items():
            res[f"{tag}.gnorm.{n}"] = np.float64(v)
        if tag == "mono18":
            with torch.no_grad():
                b2 = {k: ([x.clone() for x in v] if isinstance(v, list) else v.clone()) for k, v in batch.items()}
                b2["pose_net_input"] = torch.cat([b2["img"]] + b2["ctx_img"], 1)
                poses = model.pose_net(b2)["pose_pred"]
                res[f"{tag}.pose0"] = np_(poses[0]); res[f"{tag}.pose1"] = np_(poses[1])
    np.savez_compressed(os.path.join(OUT, "models.npz"), **res)
    print("models.npz", len(res), "arrays")


PROBE_PARAMS_PACKNET = ["depth_net.pre_calc.conv_base.weight", "depth_net.pack1.conv3d.weight", "depth_net.pack1.conv3d.bias",
                        "depth_net.pack1.conv.conv_base.weight", "depth_net.conv3.1.conv3.weight", "depth_net.conv5.2.normalize.weight",
                        "depth_net.pack5.conv.conv_base.weight", "depth_net.unpack5.conv3d.weight", "depth_net.unpack3.conv.conv_base.bias",
                        "depth_net.iconv3.conv_base.weight", "depth_net.iconv1.normalize.bias", "depth_net.disp2_layer.conv1.weight",
                        "pose_net.conv1.0.weight"]


def gen_packnet(ref):
    """MonoDepth2Model + PackNet01 (config 5 of BASELINE.json: packnet_1a.yaml, VAR_LOSS_WEIGHT 1e-4), B=1, 64x192, CPU fp32."""
    res = {}
    for tag, version in [("packnet1A", "A"), ("packnet1B", "B")]:
        sd = OM.init_packnet_state_dict(version, seed=5)
        model = ref.MonoDepth2Fixed(ref_harness.make_cfg("MonoDepth2Model", depth_net="PackNet01", version="1" + version, VAR_LOSS_WEIGHT=1e-4))
        load_ref_weights(model, sd)
        model.train()
        batch = mono_batch(1, 64, 192, 21)
        out = model({k: ([x.clone() for x in v] if isinstance(v, list) else v.clone()) for k, v in batch.items()})
        (out["rec_loss"] + out["smooth_loss"] + out["var_loss"]).backward()
        for k in ("rec_loss", "smooth_loss", "var_loss"):
            res[f"{tag}.{k}"] = np.float64(out[k].item())
        for n, v in grad_norms(model, PROBE_PARAMS_PACKNET).items():
            res[f"{tag}.gnorm.{n}"] = np.float64(v)
        model.eval()
        with torch.no_grad():
            b2 = {k: ([x.clone() for x in v] if isinstance(v, list) else v.clone()) for k, v in batch.items()}
            b2["depth_net_input"] = (b2["img"] - model.pixel_mean) / model.pixel_std
            depths = model.depth_net(b2)["depth_pred"]
            for i, d in enumerate(depths):
                res[f"{tag}.depth{i}"] = np_(d)
            fb = dict(b2); fb["flip"] = True
            res[f"{tag}.flip_depth0"] = np_(model.depth_net(fb)["depth_pred"][0])
    np.savez_compressed(os.path.join(OUT, "packnet.npz"), **res)
    print("packnet.npz", len(res), "arrays")


def gen_options(ref):
    """Config switches off in the reference's YAMLs but on its path: MODEL.DEPTH_NET.UPSAMPLE_DEPTH=True in training (DepthResNet.py:L62-63)."""
    res = {}
    sd = OM.init_state_dict(18, with_pose=True, seed=7)
    cfg = ref_harness.make_cfg("MonoDepth2Model", "18")
    cfg.MODEL.DEPTH_NET.UPSAMPLE_DEPTH = True
    model = ref.MonoDepth2Fixed(cfg)
    load_ref_weights(model, sd)
    model.train()
    batch = mono_batch(2, 64, 192, 21)
    out = model({k: ([x.clone() for x in v] if isinstance(v, list) else v.clone()) for k, v in batch.items()})
    (out["rec_loss"] + out["smooth_loss"]).backward()
    res["mono18_up.rec_loss"] = np.float64(out["rec_loss"].item()); res["mono18_up.smooth_loss"] = np.float64(out["smooth_loss"].item())
    for n, v in grad_norms(model, PROBE_PARAMS + ["pose_net.conv1.0.weight", "pose_net.pose_pred.weight"]).items():
        res[f"mono18_up.gnorm.{n}"] = np.float64(v)
    # LOSS.CLIP > 0 (MonoDepth2.py:L147-149), with both reductions
    for red in ("min", "mean"):
        tag = f"mono18_clip_{red}"
        model = ref.MonoDepth2Fixed(ref_harness.make_cfg("MonoDepth2Model", "18", CLIP=0.5, PHOTOMETRIC_REDUCE=red))
        load_ref_weights(model, sd)
        model.train()
        out = model({k: ([x.clone() for x in v] if isinstance(v, list) else v.clone()) for k, v in batch.items()})
        (out["rec_loss"] + out["smooth_loss"]).backward()
        res[f"{tag}.rec_loss"] = np.float64(out["rec_loss"].item()); res[f"{tag}.smooth_loss"] = np.float64(out["smooth_loss"].item())
        for n, v in grad_norms(model, PROBE_PARAMS + ["pose_net.conv1.0.weight", "pose_net.pose_pred.weight"]).items():
            res[f"{tag}.gnorm.{n}"] = np.float64(v)
    np.savez_compressed(os.path.join(OUT, "options.npz"), **res)
    print("options.npz", len(res), "arrays")


def gen_mono50(ref):
    """BASELINE.json configs[3]: MonoDepth2Model with the ResNet-50 encoder (projects/MonoDepth2/configs/resnet50.yaml), B=1, 64x192, CPU fp32."""
    res = {}
    tag = "mono50"
    sd = OM.init_state_dict(50, with_pose=True, seed=57)
    model = ref.MonoDepth2Fixed(ref_harness.make_cfg("MonoDepth2Model", "50"))
    load_ref_weights(model, sd)
    model.train()
    batch = mono_batch(1, 64, 192, 23)
    out = model({k: ([x.clone() for x in v] if isinstance(v, list) else v.clone()) for k, v in batch.items()})
    (out["rec_loss"] + out["smooth_loss"]).backward()
    res[f"{tag}.rec_loss"] = np.float64(out["rec_loss"].item()); res[f"{tag}.smooth_loss"] = np.float64(out["smooth_loss"].item())
    for n, v in grad_norms(model, PROBE_PARAMS_50 + ["depth_net.encoder.encoder.layer3.4.conv3.weight", "pose_net.conv1.0.weight",
                                                     "pose_net.conv7.1.weight", "pose_net.pose_pred.weight"]).items():
        res[f"{tag}.gnorm.{n}"] = np.float64(v)
    model.eval()
    with torch.no_grad():
        b2 = {k: ([x.clone() for x in v] if isinstance(v, list) else v.clone()) for k, v in batch.items()}
        res[f"{tag}.eval_depth"] = np_(model(b2)["depth_pred"])
    np.savez_compressed(os.path.join(OUT, "mono50.npz"), **res)
    print("mono50.npz", len(res), "arrays")


def main():
    os.makedirs(OUT, exist_ok=True)
    torch.set_num_threads(8)
    ref = ref_harness.load()
    if "--options-only" in sys.argv:
        gen_options(ref)
        return
    if "--mono50-only" in sys.argv:
        gen_mono50(ref)
        return
    if "--packnet-only" not in sys.argv:
        gen_geometry(ref)
        gen_models(ref)
    gen_packnet(ref)
    gen_options(ref)
    gen_mono50(ref)


if __name__ == "__main__":
    main()
