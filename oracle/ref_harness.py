"""Oracle tooling (build container only): import the reference's own Python files, unmodified.

/root/reference is NOT present on the GPU box and is never copied; this module is used only by
oracle/gen_golden.py (to write tests/golden/*.npz) and by the optional container-only
cross-checks in tests/ (skipped when /root/reference is absent).

Recipe: stub parent packages whose __path__ points at the reference
directories, so relative imports resolve while every ``__init__.py`` (which pulls fvcore / cv2 /
Google* nets) is bypassed; pre-seed small stand-ins for the two missing third-party modules.
"""
import importlib
import os
import sys
import types

import torch
import torch.nn as nn

REF_ROOT = os.environ.get("SDE_REFERENCE_ROOT", "/root/reference")
REF_PKG = os.path.join(REF_ROOT, "detectron2")


def available():
    return os.path.isdir(REF_PKG)


# --- stand-in for fvcore.common.registry.Registry (third-party, not installed) -------------------
class _Registry:
    def __init__(self, name):
        self._name, self._map = name, {}

    def register(self, obj=None):
        if obj is None:
            def deco(o):
                self._map[o.__name__] = o
                return o
            return deco
        self._map[obj.__name__] = obj
        return obj

    def get(self, name):
        if name not in self._map:
            raise KeyError(f"No object named '{name}' found in '{self._name}' registry!")
        return self._map[name]


# --- stand-in for torchvision.models (third-party 0.9.0, not installed; parity unpinned) --------
class _BasicBlock(nn.Module):
    expansion = 1

    def __init__(self, inplanes, planes, stride=1, downsample=None, groups=1, base_width=64, dilation=1,
                 norm_layer=None):
        super().__init__()
        norm_layer = norm_layer or nn.BatchNorm2d
        self.conv1 = nn.Conv2d(inplanes, planes, 3, stride, 1, bias=False)
        self.bn1 = norm_layer(planes)
        self.relu = nn.ReLU(inplace=True)
        self.conv2 = nn.Conv2d(planes, planes, 3, 1, 1, bias=False)
        self.bn2 = norm_layer(planes)
        self.downsample = downsample

    def forward(self, x):
        idt = x if self.downsample is None else self.downsample(x)
        out = self.relu(self.bn1(self.conv1(x)))
        out = self.bn2(self.conv2(out))
        return self.relu(out + idt)


class _Bottleneck(nn.Module):
    expansion = 4

    def __init__(self, inplanes, planes, stride=1, downsample=None, groups=1, base_width=64, dilation=1,
                 norm_layer=None):
        super().__init__()
        norm_layer = norm_layer or nn.BatchNorm2d
        self.conv1 = nn.Conv2d(inplanes, planes, 1, bias=False)
        self.bn1 = norm_layer(planes)
        self.conv2 = nn.Conv2d(planes, planes, 3, stride, 1, bias=False)
        self.bn2 = norm_layer(planes)
        self.conv3 = nn.Conv2d(planes, planes * 4, 1, bias=False)
        self.bn3 = norm_layer(planes * 4)
        self.relu = nn.ReLU(inplace=True)
        self.downsample = downsample

    def forward(self, x):
        idt = x if self.downsample is None else self.downsample(x)
        out = self.relu(self.bn1(self.conv1(x)))
        out = self.relu(self.bn2(self.conv2(out)))
        out = self.bn3(self.conv3(out))
        return self.relu(out + idt)


class _ResNet(nn.Module):
    def __init__(self, block, layers, num_classes=1000, norm_layer=None):
        super().__init__()
        self._norm_layer = norm_layer or nn.BatchNorm2d
        self.inplanes = 64
        self.conv1 = nn.Conv2d(3, 64, 7, 2, 3, bias=False)
        self.bn1 = self._norm_layer(64)
        self.relu = nn.ReLU(inplace=True)
        self.maxpool = nn.MaxPool2d(3, 2, 1)
        self.layer1 = self._make_layer(block, 64, layers[0])
        self.layer2 = self._make_layer(block, 128, layers[1], stride=2)
        self.layer3 = self._make_layer(block, 256, layers[2], stride=2)
        self.layer4 = self._make_layer(block, 512, layers[3], stride=2)
        self.avgpool = nn.AdaptiveAvgPool2d((1, 1))
        self.fc = nn.Linear(512 * block.expansion, num_classes)
        for m in self.modules():
            if isinstance(m, nn.Conv2d):
                nn.init.kaiming_normal_(m.weight, mode="fan_out", nonlinearity="relu")
            elif isinstance(m, (nn.BatchNorm2d, nn.GroupNorm)):
                nn.init.constant_(m.weight, 1)
                nn.init.constant_(m.bias, 0)

    def _make_layer(self, block, planes, blocks, stride=1):
        ds = None
        if stride != 1 or self.inplanes != planes * block.expansion:
            ds = nn.Sequential(nn.Conv2d(self.inplanes, planes * block.expansion, 1, stride, bias=False),
                               self._norm_layer(planes * block.expansion))
        layers = [block(self.inplanes, planes, stride, ds, norm_layer=self._norm_layer)]
        self.inplanes = planes * block.expansion
        layers += [block(self.inplanes, planes, norm_layer=self._norm_layer) for _ in range(1, blocks)]
        return nn.Sequential(*layers)


def _mk(block, layers):
    def ctor(pretrained=False, norm_layer=None, **kw):
        assert not pretrained, "no network: ImageNet weights unavailable (SURVEY 8c)"
        return _ResNet(block, layers, norm_layer=norm_layer)
    return ctor


_loaded = {}


def load():
    """Import the reference's hot-path modules; returns a namespace of them."""
    if _loaded:
        return types.SimpleNamespace(**_loaded)
    assert available(), f"reference not found at {REF_PKG}"

    def stub(name, path):
        m = types.ModuleType(name)
        m.__path__ = [path]
        sys.modules[name] = m
        return m

    stub("detectron2", REF_PKG)
    for sub in ("utils", "geometry", "layers", "modeling"):
        stub("detectron2." + sub, os.path.join(REF_PKG, sub))
    for sub in ("losses", "depth_net", "pose_net", "meta_arch"):
        stub("detectron2.modeling." + sub, os.path.join(REF_PKG, "modeling", sub))

    fv = types.ModuleType("fvcore"); fvc = types.ModuleType("fvcore.common"); fvr = types.ModuleType("fvcore.common.registry")
    fvr.Registry = _Registry
    sys.modules.setdefault("fvcore", fv); sys.modules.setdefault("fvcore.common", fvc)
    sys.modules["fvcore.common.registry"] = fvr

    tv = types.ModuleType("torchvision"); tvm = types.ModuleType("torchvision.models"); tvr = types.ModuleType("torchvision.models.resnet")
    tvr.BasicBlock, tvr.Bottleneck, tvr.ResNet, tvr.model_urls = _BasicBlock, _Bottleneck, _ResNet, {}
    tvm.ResNet, tvm.resnet = _ResNet, tvr
    tvm.resnet18 = _mk(_BasicBlock, [2, 2, 2, 2]); tvm.resnet34 = _mk(_BasicBlock, [3, 4, 6, 3])
    tvm.resnet50 = _mk(_Bottleneck, [3, 4, 6, 3]); tvm.resnet101 = _mk(_Bottleneck, [3, 4, 23, 3])
    tvm.resnet152 = _mk(_Bottleneck, [3, 8, 36, 3])
    tv.models = tvm
    sys.modules["torchvision"] = tv; sys.modules["torchvision.models"] = tvm; sys.modules["torchvision.models.resnet"] = tvr

    imp = importlib.import_module
    _loaded["camera"] = imp("detectron2.geometry.camera")
    _loaded["pose_utils"] = imp("detectron2.geometry.pose_utils")
    _loaded["ssim_loss"] = imp("detectron2.modeling.losses.ssim_loss")
    _loaded["smoothness_loss"] = imp("detectron2.modeling.losses.smoothness_loss")
    _loaded["losses"] = imp("detectron2.modeling.losses.losses")
    _loaded["depth_decoder"] = imp("detectron2.layers.depth_decoder")
    _loaded["resnet_encoder"] = imp("detectron2.layers.resnet_encoder")
    dn_build = imp("detectron2.modeling.depth_net.build")
    _loaded["DepthResNet"] = imp("detectron2.modeling.depth_net.DepthResNet")
    _loaded["layers01"] = imp("detectron2.layers.layers01")
    _loaded["PackNet01"] = imp("detectron2.modeling.depth_net.PackNet01")
    sys.modules["detectron2.modeling.depth_net"].build_depth_net = dn_build.build_depth_net
    pn_build = imp("detectron2.modeling.pose_net.build")
    _loaded["PoseNet"] = imp("detectron2.modeling.pose_net.PoseNet")
    sys.modules["detectron2.modeling.pose_net"].build_pose_net = pn_build.build_pose_net
    _loaded["meta_build"] = imp("detectron2.modeling.meta_arch.build")
    _loaded["MonoDepth2"] = imp("detectron2.modeling.meta_arch.MonoDepth2")
    _loaded["Supervised"] = imp("detectron2.modeling.meta_arch.Supervised")

    base = _loaded["MonoDepth2"].MonoDepth2Model

    class MonoDepth2Fixed(base):
        """Intended semantics: broadcast t to [B,3,h,w] before view_synthesis."""

        def rgb_consistency_loss(self, A, B, D, K, R=None, t=None):
            t = None if t is None else t.expand(-1, -1, *D.shape[-2:])
            return super().rgb_consistency_loss(A, B, D, K, R, t)

    _loaded["MonoDepth2Fixed"] = MonoDepth2Fixed
    return types.SimpleNamespace(**_loaded)


def make_cfg(meta_arch, encoder="18", device="cpu", depth_net="DepthResNet", version="1A", **loss_over):
    """Nested SimpleNamespace carrying exactly the cfg keys the path reads."""
    NS = types.SimpleNamespace
    loss = dict(SSIM_WEIGHT=0.85, C1=1e-4, C2=9e-4, CLIP=0.0, AUTOMASK=True, SMOOTHNESS_WEIGHT=1e-3,
                PHOTOMETRIC_REDUCE="min", SUPERVISED_WEIGHT=0.0, VARIANCE_FOCUS=0.85, VAR_LOSS_WEIGHT=0.0)
    loss.update(loss_over)
    return NS(MODEL=NS(META_ARCHITECTURE=meta_arch, DEVICE=device, MAX_DEPTH=80,
                       PIXEL_MEAN=[0.485, 0.456, 0.406], PIXEL_STD=[0.229, 0.224, 0.225],
                       DEPTH_NET=NS(NAME=depth_net, ENCODER_NAME=encoder, UPSAMPLE_DEPTH=False, VERSION=version),
                       POSE_NET=NS(NAME="PoseNet", NUM_CONTEXTS=2)),
              LOSS=NS(**loss))
