"""Meta-architecture registry and build_model (reference: detectron2/modeling/meta_arch/build.py:L6-23)."""
import torch

from ...utils.registry import Registry

META_ARCH_REGISTRY = Registry("META_ARCH")
META_ARCH_REGISTRY.__doc__ = "Registry for meta-architectures, i.e. the whole model: obj(cfg) -> nn.Module."


def build_model(cfg):
    """Build the whole model architecture, defined by ``cfg.MODEL.META_ARCHITECTURE`` (does not load weights)."""
    meta_arch = cfg.MODEL.META_ARCHITECTURE
    model = META_ARCH_REGISTRY.get(meta_arch)(cfg)
    model.to(torch.device(cfg.MODEL.DEVICE))
    return model
