"""MonoDepth2Model (reference: detectron2/modeling/meta_arch/MonoDepth2.py:L21-151), intended semantics of SURVEY.md fact 4
(the translation is constant per sample).  The per-(scale, context) chain view_synthesis -> SSIM -> L1 mix -> min/auto-mask
is one fused kernel per scale; the image pyramid is built once per batch instead of once per (scale, context)."""
from collections import defaultdict

import torch

from ...hip import lib as L
from ...hip import photometric as HP
from ...utils.memory import to_cuda
from ..losses.losses import silog_loss, variance_loss
from ..losses.smoothness_loss import smoothness_loss
from ..losses.ssim_loss import SSIM
from ..pose_net import build_pose_net
from .build import META_ARCH_REGISTRY
from .common import HipMetaArch


# Every scale of the photometric and the smoothness term in one launch per phase (sde_mono_loss_fwd / _bwd; bench.py --opt photo_multi=0 selects the
# per-scale path).  Kernel time: photometric forward 116 us against 90 + 31 + 19 + 14, backward 187 against 148 + 50 + 22 + 17 (192x640, bs 12), and ~30 fewer
# launch-sized kernels on the serial seam between the two passes.  With the pose-gradient sum enqueued on PoseNet's stream first (hip/photometric.py) the
# replayed step is faster too: MonoDepth2-R18 4.33 -> 4.15 ms, R50 7.69 -> 7.48 (photometric launches alone; before that change the runtime's queue
# assignment put PoseNet's backward behind the depth network's and the step was SLOWER, 4.46 / 7.83).
MULTI_SCALE_PHOTO = True


@META_ARCH_REGISTRY.register()
class MonoDepth2Model(HipMetaArch):
    def __init__(self, cfg):
        HipMetaArch.__init__(self, cfg)
        self.pose_net = build_pose_net(cfg)
        loss = cfg.LOSS
        self.ssim = SSIM(loss.C1, loss.C2)
        self.ssim_loss_weight, self.photometric_reduce, self.use_automask, self.clip_loss = loss.SSIM_WEIGHT, loss.PHOTOMETRIC_REDUCE, loss.AUTOMASK, loss.CLIP
        self.var_loss_w, self.sup_loss_w, self.smooth_loss_w = loss.VAR_LOSS_WEIGHT, loss.SUPERVISED_WEIGHT, loss.SMOOTHNESS_WEIGHT
        self.supervise_loss = silog_loss(loss.VARIANCE_FOCUS)

    def _image_pyramid(self, batch):
        """[(target image, [context images]) resized to each prediction scale]: the scales are those of the depth decoder, (H, W) >> i."""
        image, contexts = batch["img_orig"], batch["ctx_img_orig"]
        H, W = image.shape[-2:]
        n = len(getattr(self.depth_net, "scales", range(4)))
        if getattr(self.depth_net, "upsample_depth", False):
            return None
        return [(HP.resize(image, (H >> i, W >> i)), [HP.resize(c, (H >> i, W >> i)) for c in contexts]) for i in range(n)]

    def forward(self, batch):
        output = {}
        if not self.training:
            batch = self.run_depth_net(batch)
            output["depth_pred"] = batch["depth_pred"][0]
            return output
        batch = to_cuda(batch, self.device)
        pose_stream = None
        if L.POSE_STREAM and self.device.type == "cuda":
            # PoseNet underneath the depth network: its kernels are launch-bound (seven small layers), the depth network's are not; autograd
            # replays its backward on the same auxiliary stream, i.e. underneath the depth network's backward
            cur, aux = torch.cuda.current_stream(), L.aux_stream()
            aux.wait_stream(cur)
            with torch.cuda.stream(aux):
                batch["pose_net_input"] = torch.cat([batch["img"]] + batch["ctx_img"], 1)   # augmented frames, not normalised (L65)
                batch = self.pose_net(batch)
                pyramid = self._image_pyramid(batch)                                         # needs neither network: off the main stream as well
                L.mark("pose_fwd_end")
            L.AUX_USED = True
            pose_stream = aux
            batch = self.run_depth_net(batch)
            L.mark("depth_fwd_end")
            cur.wait_stream(aux)
        else:
            batch = self.run_depth_net(batch)
            batch["pose_net_input"] = torch.cat([batch["img"]] + batch["ctx_img"], 1)       # augmented frames, not normalised (L65)
            batch = self.pose_net(batch)
            pyramid = None
        image, contexts, intrinsics = batch["img_orig"], batch["ctx_img_orig"], batch["intrinsics"].float().contiguous()
        depth_pred, poses = batch["depth_pred"], batch["pose_pred"]
        num_scales = len(depth_pred)
        if L.MARKS is not None:      # diagnostic timeline: when the two networks' backward passes receive their first gradients
            def _marker(name):
                def hook(g):
                    L.mark(name)
                return hook
            depth_pred[0].register_hook(_marker("depth_bwd_start"))
            poses[0].register_hook(_marker("pose_bwd_start"))
        H, W = image.shape[-2:]
        terms = defaultdict(lambda: ([], []))        # loss name -> (per-scale 0-d tensors, their weights)
        photo_losses, pyr, smooth_ws = [], [], []
        # every scale of the photometric loss in ONE launch per phase (sde_photo_multi_fwd / _bwd) unless LOSS.CLIP needs the per-scale statistics
        multi = MULTI_SCALE_PHOTO and not (self.clip_loss and self.clip_loss > 0.0) and 1 <= num_scales <= HP.PH_MAX_SCALES
        for i in range(num_scales):
            scale_w = 1.0 / 2 ** (num_scales - i - 1)
            h, w = depth_pred[i].shape[-2:]
            if pyramid is not None and i < len(pyramid) and pyramid[i][0].shape[-2:] == (h, w):
                resized_image, resized_targets = pyramid[i]
            else:
                resized_image = HP.resize(image, (h, w))
                resized_targets = [HP.resize(c, (h, w)) for c in contexts]
            pyr.append((resized_image, resized_targets, w / W, h / H))
            if not multi:
                photo_losses.append(HP.photometric_scale_loss(depth_pred[i], intrinsics, resized_image, resized_targets, poses, w / W, h / H,
                                                              ssim_w=self.ssim_loss_weight, C1=self.ssim.C1, C2=self.ssim.C2,
                                                              automask=self.use_automask, reduce=self.photometric_reduce, clip=self.clip_loss))
            def add(name, value, weight):
                terms[name][0].append(value); terms[name][1].append(weight)
            if self.smooth_loss_w > 0.0:
                if multi:
                    smooth_ws.append(scale_w * self.smooth_loss_w / num_scales)       # the term itself rides in the all-scales launches below
                else:
                    add("smooth_loss", smoothness_loss(depth_pred[i], resized_image), scale_w * self.smooth_loss_w / num_scales)
            if self.sup_loss_w > 0.0:
                # the reference weights this term with smooth_loss_w (MonoDepth2.py:L109, sic)
                add("sup_loss", self.supervise_loss(depth_pred[i], batch["depth"]), scale_w * self.smooth_loss_w / num_scales)
            if self.var_loss_w > 0.0:
                add("var_loss", variance_loss(depth_pred[i]), scale_w * self.var_loss_w / num_scales)
        # the reference accumulates `loss += term_i * w_i` scale by scale (MonoDepth2.py:L103-112, L126): per loss that is 2 tiny kernels per scale forward
        # and as many backward; one stack + one dot product with a cached weight vector is the same sum (fp32, 4 terms) in 2 + 1 kernels
        if multi:
            # photometric + smoothness terms of all scales, weighted and summed, in four launches forward and two backward (sde_mono_loss_fwd / _bwd)
            rec, smooth, _ = HP.mono_loss(depth_pred, intrinsics, [p[0] for p in pyr], [p[1] for p in pyr], poses, [(p[2], p[3]) for p in pyr],
                                          [1.0 / num_scales] * num_scales, smooth_ws or None, ssim_w=self.ssim_loss_weight, C1=self.ssim.C1, C2=self.ssim.C2,
                                          automask=self.use_automask, reduce=self.photometric_reduce, pose_stream=pose_stream)
            output["rec_loss"] = rec
            if smooth_ws:
                output["smooth_loss"] = smooth
        else:
            output["rec_loss"] = self._weighted_sum(photo_losses, [1.0 / num_scales] * num_scales)
        for name, (vals, ws) in terms.items():
            output[name] = self._weighted_sum(vals, ws)
        return output
