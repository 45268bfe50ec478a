"""PoseNet: 7 x (stride-2 conv + GroupNorm(16) + ReLU), 1x1 head, global mean, x0.01, pose_vec2mat
(reference: detectron2/modeling/pose_net/PoseNet.py:L13-65).  ``convN`` are Sequential(conv, groupnorm) so the reference's
state-dict keys ``pose_net.convN.0.*`` / ``pose_net.convN.1.*`` are preserved (its ReLU at index 2 has no parameters)."""
import torch
import torch.nn as nn

from ...hip import nn as HN
from ...hip import photometric as HP
from ...layers.hip_modules import HipConv2d, HipGroupNorm
from ..depth_net.DepthResNet import compute_dtype
from .build import POSE_NET_REGISTRY


def conv_gn_relu(in_planes, out_planes, kernel_size=3, stride=2):
    return nn.Sequential(HipConv2d(in_planes, out_planes, kernel_size, stride, (kernel_size - 1) // 2, bias=True), HipGroupNorm(16, out_planes))


@POSE_NET_REGISTRY.register()
class PoseNet(nn.Module):
    def __init__(self, cfg, **kwargs):
        super().__init__()
        self.nb_ref_imgs = cfg.MODEL.POSE_NET.NUM_CONTEXTS
        channels = [16, 32, 64, 128, 256, 256, 256]
        self.conv1 = conv_gn_relu(3 * (1 + self.nb_ref_imgs), channels[0], kernel_size=7)
        self.conv2 = conv_gn_relu(channels[0], channels[1], kernel_size=5)
        self.conv3 = conv_gn_relu(channels[1], channels[2])
        self.conv4 = conv_gn_relu(channels[2], channels[3])
        self.conv5 = conv_gn_relu(channels[3], channels[4])
        self.conv6 = conv_gn_relu(channels[4], channels[5])
        self.conv7 = conv_gn_relu(channels[5], channels[6])
        self.pose_pred = HipConv2d(channels[6], 6 * self.nb_ref_imgs, 1, 1, 0, bias=True)
        self.dtype = compute_dtype(cfg)
        self.init_weights()

    def init_weights(self):
        for m in self.modules():
            if isinstance(m, HipConv2d):
                nn.init.xavier_uniform_(m.weight.data)
                if m.bias is not None:
                    m.bias.data.zero_()

    def forward(self, batch):
        x = batch.get("pose_net_input_nhwc")
        if x is None:
            x = HN.prep_input(batch["pose_net_input"], None, None, self.dtype)
        for blk in (self.conv1, self.conv2, self.conv3, self.conv4, self.conv5, self.conv6, self.conv7):
            x = blk[1](blk[0](x))
        p = self.pose_pred(x)                                                # [B, h, w, pad(6*n)]
        n6 = 6 * self.nb_ref_imgs
        pose = p[..., :n6].float().mean(dim=(1, 2))                          # tiny [B,h,w,12] reduction: torch glue
        pose = 0.01 * pose.view(pose.size(0), self.nb_ref_imgs, 6)
        batch["pose_vec"] = pose
        batch["pose_pred"] = [HP.pose_vec2mat(pose[:, i].contiguous()) for i in range(pose.shape[1])]
        return batch
