"""ResNet-18/34/50 encoder on the HIP convolution engine.

Reference: detectron2/layers/resnet_encoder.py:L61-99 (ResnetEncoder: 5 features = relu(bn1(conv1)), layer1(maxpool), layer2-4)
wrapping torchvision 0.9 resnetXX (v1.5: stride on the 3x3 of a Bottleneck; kaiming-normal fan_out convs, BN gamma=1/beta=0).
Attribute / state-dict names follow torchvision (``encoder.conv1``, ``encoder.layer1.0.bn2``, ``...downsample.0``, ``encoder.fc``).
"""
import numpy as np
import torch
import torch.nn as nn

from ..hip import nn as HN
from .hip_modules import HipBatchNorm2d, HipConv2d, conv_bn


class BasicBlock(nn.Module):
    expansion = 1

    def __init__(self, inplanes, planes, stride=1, downsample=None):
        super().__init__()
        self.conv1 = HipConv2d(inplanes, planes, 3, stride, 1, bias=False)
        self.bn1 = HipBatchNorm2d(planes)
        self.conv2 = HipConv2d(planes, planes, 3, 1, 1, bias=False)
        self.bn2 = HipBatchNorm2d(planes)
        self.downsample = downsample

    def forward(self, x, n_out=1):
        xa, xb = x if isinstance(x, tuple) else (x, x)       # two aliases of the block input: one per consumer (see hip.nn._BatchNormAct)
        idt = xb if self.downsample is None else conv_bn(self.downsample[0], self.downsample[1], xb, relu=False)
        out = conv_bn(self.conv1, self.bn1, xa)
        return conv_bn(self.conv2, self.bn2, out, residual=idt, relu=True, n_out=n_out)


class Bottleneck(nn.Module):
    expansion = 4

    def __init__(self, inplanes, planes, stride=1, downsample=None):
        super().__init__()
        self.conv1 = HipConv2d(inplanes, planes, 1, 1, 0, bias=False)
        self.bn1 = HipBatchNorm2d(planes)
        self.conv2 = HipConv2d(planes, planes, 3, stride, 1, bias=False)
        self.bn2 = HipBatchNorm2d(planes)
        self.conv3 = HipConv2d(planes, planes * 4, 1, 1, 0, bias=False)
        self.bn3 = HipBatchNorm2d(planes * 4)
        self.downsample = downsample

    def forward(self, x, n_out=1):
        xa, xb = x if isinstance(x, tuple) else (x, x)       # two aliases of the block input: one per consumer (see hip.nn._BatchNormAct)
        idt = xb if self.downsample is None else conv_bn(self.downsample[0], self.downsample[1], xb, relu=False)
        out = conv_bn(self.conv1, self.bn1, xa)
        out = conv_bn(self.conv2, self.bn2, out)
        return conv_bn(self.conv3, self.bn3, out, residual=idt, relu=True, n_out=n_out)


class ResNet(nn.Module):
    """torchvision-shaped container (conv1, bn1, layer1..4, fc); ``fc`` is kept only so checkpoints load."""

    def __init__(self, block, layers, num_classes=1000):
        super().__init__()
        self.inplanes = 64
        self.conv1 = HipConv2d(3, 64, 7, 2, 3, bias=False)
        self.bn1 = HipBatchNorm2d(64)
        self.layer1 = self._make_layer(block, 64, layers[0])
        self.layer2 = self._make_layer(block, 128, layers[1], stride=2)
        self.layer3 = self._make_layer(block, 256, layers[2], stride=2)
        self.layer4 = self._make_layer(block, 512, layers[3], stride=2)
        self.fc = nn.Linear(512 * block.expansion, num_classes)
        for m in self.modules():
            if isinstance(m, HipConv2d):
                nn.init.kaiming_normal_(m.weight, mode="fan_out", nonlinearity="relu")

    def _make_layer(self, block, planes, blocks, stride=1):
        downsample = None
        if stride != 1 or self.inplanes != planes * block.expansion:
            downsample = nn.Sequential(HipConv2d(self.inplanes, planes * block.expansion, 1, stride, 0, bias=False),
                                       HipBatchNorm2d(planes * block.expansion))
        layers = [block(self.inplanes, planes, stride, downsample)]
        self.inplanes = planes * block.expansion
        layers += [block(self.inplanes, planes) for _ in range(1, blocks)]
        return nn.Sequential(*layers)


_SPECS = {18: (BasicBlock, [2, 2, 2, 2]), 34: (BasicBlock, [3, 4, 6, 3]), 50: (Bottleneck, [3, 4, 6, 3])}


class ResnetEncoder(nn.Module):
    def __init__(self, num_layers, pretrained=False, num_input_images=1, norm_layer=None):
        super().__init__()
        if num_layers not in _SPECS:
            raise ValueError("{} is not a valid number of resnet layers".format(num_layers))
        if pretrained:
            raise RuntimeError("ImageNet weights cannot be downloaded here; load them from a local checkpoint with load_state_dict "
                               "(use ENCODER_NAME '18'/'50' instead of '18pt'/'50pt')")
        if num_input_images != 1 or norm_layer is not None:
            raise NotImplementedError("multi-image / custom-norm encoders are not on the path")
        self.num_ch_enc = np.array([64, 64, 128, 256, 512])
        block, layers = _SPECS[num_layers]
        self.encoder = ResNet(block, layers)
        if num_layers > 34:
            self.num_ch_enc[1:] *= 4

    def forward(self, x, cut=None):
        """x: NHWC normalised image (channels padded).  Returns the 5 NHWC feature maps.
        cut (optional): callable(list_of_3_features) -> list of 3 detached leaves; layer3/layer4 then consume the leaves, so a
        backward pass from the loss stops there (HipTrainer's two-phase backward)."""
        e = self.encoder
        split = torch.is_grad_enabled() and cut is None      # aliases only matter for backward; with a cut the features become leaves

        def run(layer, x, last_n):
            """The blocks of one layer; every block output but the last gets two aliases (conv1 + residual / down-sampling of the next block),
            the last one `last_n` (next layer's two consumers + the decoder skip)."""
            blocks = list(layer)
            for i, blk in enumerate(blocks):
                n = (2 if i + 1 < len(blocks) else last_n) if split else 1
                x = blk(x, n_out=n)
            return x

        if split:
            f0, f0_pool = conv_bn(e.conv1, e.bn1, x, n_out=2)            # decoder skip + max-pool
            o1 = run(e.layer1, HN.max_pool_3x3_s2(f0_pool, n_out=2), 3)    # first block: convolution + residual / down-sampling path
            o2 = run(e.layer2, (o1[1], o1[2]), 3)
            o3 = run(e.layer3, (o2[1], o2[2]), 3)
            f4 = run(e.layer4, (o3[1], o3[2]), 1)
            self.features = [f0, o1[0], o2[0], o3[0], f4]
            return self.features
        f0 = conv_bn(e.conv1, e.bn1, x)
        f1 = run(e.layer1, HN.max_pool_3x3_s2(f0), 1)
        f2 = run(e.layer2, f1, 1)
        if cut is not None:
            f0, f1, f2 = cut([f0, f1, f2])
        f3 = run(e.layer3, f2, 1)
        f4 = run(e.layer4, f3, 1)
        self.features = [f0, f1, f2, f3, f4]
        return self.features
