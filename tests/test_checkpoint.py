"""CPU: checkpoint save / load / resume in the reference's file format (simpledepthestimation_amd.checkpoint).

fvcore's Checkpointer (what the reference subclasses) is absent here, so the pins are (1) the file layout the reference's call sites rely on
(projects/*/train.py: {"model", "optimizer"[, "scheduler"], "iteration"}, model_{i:07d}.pth, model_final.pth, last_checkpoint), (2) torch's own
optimizer: a state dict written by torch.optim.AdamW over the same parameter groups must restore the trainer to the same trajectory and vice
versa, and (3) the reference's state-dict keys as restated by the oracle (which is pinned to the reference by tests/golden)."""
import os

import pytest
import torch
import torch.nn as nn

from simpledepthestimation_amd.checkpoint import DetectionCheckpointer, PeriodicCheckpointer
from simpledepthestimation_amd.engine.trainer import HipTrainer, ParamGroup
from test_dp_gloo import Tiny, torch_adam


def _trainer(model):
    groups = [ParamGroup("a", model.a.named_parameters(prefix="a"), 1e-2, 1e-2), ParamGroup("b", model.b.named_parameters(prefix="b"), 5e-3, 0.0)]
    return HipTrainer(model, groups, adamw=True, eps=1e-6, adam_fn=torch_adam)


def _batch(seed):
    g = torch.Generator().manual_seed(seed)
    return {"x": torch.randn(8, 6, generator=g), "t": torch.randn(8, generator=g)}


def _torch_reference(steps):
    torch.manual_seed(3)
    model = Tiny()
    opt = torch.optim.AdamW([{"params": list(model.a.parameters()), "lr": 1e-2, "weight_decay": 1e-2},
                             {"params": list(model.b.parameters()), "lr": 5e-3, "weight_decay": 0.0}], eps=1e-6)
    for i in range(steps):
        opt.zero_grad()
        model(_batch(i))["mse_loss"].backward()
        opt.step()
    return model, opt


def test_save_layout_and_resume_continues_the_trajectory(tmp_path):
    torch.manual_seed(3)
    model = Tiny(); tr = _trainer(model)
    for i in range(3):
        tr.step(_batch(i))
    ck = DetectionCheckpointer(model, str(tmp_path), optimizer=tr)
    per = PeriodicCheckpointer(ck, 1, max_iter=10)
    per.step(2)
    assert sorted(os.listdir(tmp_path)) == ["last_checkpoint", "model_0000002.pth"]
    assert open(tmp_path / "last_checkpoint").read() == "model_0000002.pth"
    raw = torch.load(tmp_path / "model_0000002.pth", weights_only=True)
    assert set(raw) == {"model", "optimizer", "iteration"} and raw["iteration"] == 2
    assert set(raw["model"]) == set(model.state_dict())
    assert set(raw["optimizer"]) == {"state", "param_groups"} and set(raw["optimizer"]["state"][0]) == {"step", "exp_avg", "exp_avg_sq"}
    for i in range(3, 5):
        tr.step(_batch(i))
    # a fresh process: different init, resume from the directory
    torch.manual_seed(77)
    model2 = Tiny(); tr2 = _trainer(model2)
    ck2 = DetectionCheckpointer(model2, str(tmp_path), optimizer=tr2)
    assert ck2.has_checkpoint()
    rest = ck2.resume_or_load("", resume=True)
    assert rest == {"iteration": 2} and tr2.t == 3
    for i in range(3, 5):
        tr2.step(_batch(i))
    for (k, a), (_, b) in zip(model.state_dict().items(), model2.state_dict().items()):
        assert torch.equal(a, b), k
    assert torch.equal(tr.m, tr2.m) and torch.equal(tr.v, tr2.v)


def test_resume_false_loads_weights_only(tmp_path):
    torch.manual_seed(3)
    model = Tiny(); tr = _trainer(model)
    tr.step(_batch(0))
    path = DetectionCheckpointer(model, str(tmp_path), optimizer=tr).save("model_final", epoch=4)
    model2 = Tiny(); tr2 = _trainer(model2)
    rest = DetectionCheckpointer(model2, "", optimizer=tr2).resume_or_load(path, resume=False)
    assert rest["epoch"] == 4 and "optimizer" in rest            # not consumed: weights only
    assert tr2.t == 0 and float(tr2.m.abs().sum()) == 0.0
    assert all(torch.equal(a, b) for a, b in zip(model.state_dict().values(), model2.state_dict().values()))
    assert DetectionCheckpointer(model2, "").resume_or_load("", resume=True) == {}
    with pytest.raises(FileNotFoundError):
        DetectionCheckpointer(model2, "").load(str(tmp_path / "nope.pth"))


def test_optimizer_state_interoperates_with_torch_adamw(tmp_path):
    """A checkpoint written by the reference's optimizer class restores the fused trainer to the same trajectory, and the other way round."""
    ref_model, ref_opt = _torch_reference(3)
    torch.save({"model": ref_model.state_dict(), "optimizer": ref_opt.state_dict(), "iteration": 2}, tmp_path / "ref.pth")
    model = Tiny(); tr = _trainer(model)
    rest = DetectionCheckpointer(model, str(tmp_path), optimizer=tr).load(str(tmp_path / "ref.pth"))
    assert rest == {"iteration": 2} and tr.t == 3
    for i in range(3, 6):
        tr.step(_batch(i))
        ref_opt.zero_grad(); ref_model(_batch(i))["mse_loss"].backward(); ref_opt.step()
    for (k, a), (_, b) in zip(model.state_dict().items(), ref_model.state_dict().items()):
        torch.testing.assert_close(a, b, rtol=1e-5, atol=1e-6, msg=k)
    # trainer -> torch.optim
    ref2, opt2 = _torch_reference(0)
    ref2.load_state_dict(model.state_dict())
    opt2.load_state_dict(tr.state_dict())
    tr.step(_batch(9))
    opt2.zero_grad(); ref2(_batch(9))["mse_loss"].backward(); opt2.step()
    for (k, a), (_, b) in zip(model.state_dict().items(), ref2.state_dict().items()):
        torch.testing.assert_close(a, b, rtol=1e-5, atol=1e-6, msg=k)


def test_optimizer_state_rejects_mismatches():
    model = Tiny(); tr = _trainer(model)
    tr.step(_batch(0))
    sd = tr.state_dict()
    bad = {"state": sd["state"], "param_groups": sd["param_groups"][:1]}
    with pytest.raises(ValueError):
        tr.load_state_dict(bad)
    sd["state"][0]["exp_avg"] = torch.zeros(2, 2)
    with pytest.raises(ValueError):
        tr.load_state_dict(sd)


def test_load_tolerances_of_the_reference(tmp_path):
    """DDP 'module.' prefix, a bare state dict, shape mismatches, unknown keys, missing pixel_mean/std (checkpoint.py:L24-44)."""
    class WithPixels(Tiny):
        def __init__(self):
            super().__init__()
            self.register_buffer("pixel_mean", torch.zeros(3, 1, 1))
            self.register_buffer("pixel_std", torch.ones(3, 1, 1))
    torch.manual_seed(5)
    src = Tiny()
    sd = {"module." + k: v for k, v in src.state_dict().items()}
    sd["module.b.weight"] = torch.zeros(7, 7)                      # wrong shape: dropped, reported
    sd["module.encoder.fc.weight"] = torch.zeros(2)               # the reference's unused classifier: unexpected, tolerated
    torch.save(sd, tmp_path / "bare.pth")                          # no "model" key
    dst = WithPixels()
    before_b = dst.b.weight.detach().clone()
    ck = DetectionCheckpointer(dst, "")
    assert ck.load(str(tmp_path / "bare.pth")) == {}
    inc = ck.last_incompatible
    assert inc.missing_keys == ["b.weight"] and inc.unexpected_keys == ["encoder.fc.weight"]
    assert inc.incorrect_shapes == [("b.weight", (7, 7), (1, 5))]
    assert torch.equal(dst.a.weight, src.a.weight) and torch.equal(dst.b.weight, before_b)


def test_periodic_checkpointer_names_and_pruning(tmp_path):
    model = Tiny()
    per = PeriodicCheckpointer(DetectionCheckpointer(model, str(tmp_path)), period=2, max_iter=7, max_to_keep=2)
    for it in range(7):
        per.step(it)
    files = sorted(f for f in os.listdir(tmp_path) if f.endswith(".pth"))
    assert files == ["model_0000003.pth", "model_0000005.pth", "model_final.pth"]
    assert open(tmp_path / "last_checkpoint").read() == "model_final.pth"
    assert torch.load(tmp_path / "model_final.pth", weights_only=True)["iteration"] == 6


@pytest.mark.parametrize("arch,layers,pose", [("SupDepthModel", 50, False), ("MonoDepth2Model", 18, True)])
def test_reference_keyed_checkpoint_loads_into_the_hip_models(tmp_path, arch, layers, pose):
    """A file with the reference's state-dict keys (incl. the torchvision ``fc`` the reference carries) loads with nothing missing or left over."""
    from oracle import models as OM
    from simpledepthestimation_amd.config import get_cfg
    from simpledepthestimation_amd.modeling import build_model
    sd = OM.init_state_dict(layers, with_pose=pose, seed=4)
    sd = {k: torch.as_tensor(v) for k, v in sd.items()}
    sd["depth_net.encoder.encoder.fc.weight"] = torch.zeros(1000, 512 if layers == 18 else 2048)
    sd["depth_net.encoder.encoder.fc.bias"] = torch.zeros(1000)
    torch.save({"model": sd, "iteration": 11}, tmp_path / "model_0000011.pth")
    cfg = get_cfg(); cfg.MODEL.META_ARCHITECTURE = arch; cfg.MODEL.DEVICE = "cpu"; cfg.MODEL.DEPTH_NET.ENCODER_NAME = str(layers)
    model = build_model(cfg)
    ck = DetectionCheckpointer(model, "")
    assert ck.load(str(tmp_path / "model_0000011.pth")) == {"iteration": 11}
    assert ck.last_incompatible.missing_keys == [] and ck.last_incompatible.incorrect_shapes == []
    assert ck.last_incompatible.unexpected_keys == []          # ``fc`` is kept as an untrained module exactly so that these files load
    for k, v in model.state_dict().items():
        if k in sd:
            assert torch.equal(v.cpu(), sd[k].to(v.dtype)), k


class WithFc(nn.Module):
    """Mirrors the reference's MonoDepth2 'Depth' group (projects/MonoDepth2/train.py:L51-53: depth_net.parameters()): torchvision's unused
    classifier sits in the MIDDLE of the group (encoder ... encoder.fc, decoder ...), never receives a gradient, and the trainer skips it."""

    def __init__(self):
        super().__init__()
        self.depth_net = nn.Module()
        self.depth_net.encoder = nn.Module()
        self.depth_net.encoder.conv = nn.Linear(6, 5)
        self.depth_net.encoder.fc = nn.Linear(5, 3)           # unused, like depth_net.encoder.encoder.fc
        self.depth_net.decoder = nn.Module()
        self.depth_net.decoder.a = nn.Linear(5, 5)            # same shape as ...
        self.depth_net.decoder.b = nn.Linear(5, 5)            # ... its neighbour: a positional shift would go unnoticed by a shape check
        self.pose_net = nn.Linear(5, 1)

    def forward(self, batch):
        h = torch.tanh(self.depth_net.encoder.conv(batch["x"]))
        h = torch.tanh(self.depth_net.decoder.b(torch.tanh(self.depth_net.decoder.a(h))))
        return {"mse_loss": ((self.pose_net(h).squeeze(-1) - batch["t"]) ** 2).mean()}


def _fc_trainer(model):
    groups = [ParamGroup("Depth", model.depth_net.named_parameters(prefix="depth_net"), 2e-3, 0.0),
              ParamGroup("Pose", model.pose_net.named_parameters(prefix="pose_net"), 1e-3, 0.0)]
    return HipTrainer(model, groups, adamw=False, eps=1e-8, adam_fn=torch_adam)


def test_reference_optimizer_state_with_unused_fc_in_the_middle_of_a_group(tmp_path):
    """torch.optim.Adam over depth_net.parameters() + pose_net.parameters() (the reference's MonoDepth2 optimizer): fc.weight / fc.bias occupy
    positions 2 and 3 of the 'Depth' group and have no state.  The trainer must map every later parameter by its position in the FULL
    registration order and continue torch's trajectory exactly."""
    torch.manual_seed(11)
    ref = WithFc()
    opt = torch.optim.Adam([{"params": list(ref.depth_net.parameters()), "lr": 2e-3, "weight_decay": 0.0},
                            {"params": list(ref.pose_net.parameters()), "lr": 1e-3, "weight_decay": 0.0}])
    for i in range(3):
        opt.zero_grad(); ref(_batch(i))["mse_loss"].backward(); opt.step()
    osd = opt.state_dict()
    assert len(osd["param_groups"][0]["params"]) == 8 and 2 not in osd["state"] and 3 not in osd["state"]      # fc holds positions, not state
    torch.manual_seed(99)
    model = WithFc(); model.load_state_dict(ref.state_dict())
    tr = _fc_trainer(model)
    assert [n for n, _ in tr.groups[0].named_params] == ["depth_net.encoder.conv.weight", "depth_net.encoder.conv.bias", "depth_net.decoder.a.weight",
                                                        "depth_net.decoder.a.bias", "depth_net.decoder.b.weight", "depth_net.decoder.b.bias"]
    tr.load_state_dict(osd)
    assert tr.t == 3
    # decoder.a's moments are torch's entry 4 (not 2, which a count over the kept parameters only would pick)
    off = tr._off[id(tr.groups[0].named_params[2][1])]
    assert off % 4 == 0 and torch.equal(tr.m[off:off + 25].view(5, 5), osd["state"][4]["exp_avg"])
    for i in range(3, 6):
        tr.step(_batch(i))
        opt.zero_grad(); ref(_batch(i))["mse_loss"].backward(); opt.step()
    for (k, a), (_, b) in zip(model.state_dict().items(), ref.state_dict().items()):
        torch.testing.assert_close(a, b, rtol=1e-5, atol=1e-6, msg=k)


def test_optimizer_state_is_validated_before_anything_is_written():
    torch.manual_seed(2)
    model = WithFc(); tr = _fc_trainer(model)
    for i in range(2):
        tr.step(_batch(i))
    m0, v0, t0 = tr.m.clone(), tr.v.clone(), tr.t
    sd = tr.state_dict()
    last = max(sd["state"])
    sd["state"][last]["exp_avg_sq"] = torch.zeros(3, 3)          # wrong shape at the very end
    with pytest.raises(ValueError):
        tr.load_state_dict(sd)
    assert torch.equal(tr.m, m0) and torch.equal(tr.v, v0) and tr.t == t0      # nothing half-loaded
    sd = tr.state_dict()
    sd["param_groups"][0].pop("param_names")
    sd["param_groups"][0]["params"] = sd["param_groups"][0]["params"][:-1]      # neither the full nor the trained count
    with pytest.raises(ValueError):
        tr.load_state_dict(sd)
    assert torch.equal(tr.m, m0) and torch.equal(tr.v, v0)
