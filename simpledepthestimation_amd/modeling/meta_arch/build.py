"""``build_model(cfg)``: the drop-in entry point of the path (contract of detectron2/modeling/meta_arch/build.py:L6-23).

``cfg.MODEL.META_ARCHITECTURE`` names a class registered in META_ARCH_REGISTRY (``SupDepthModel``, ``MonoDepth2Model``); the class is
instantiated with the cfg and moved to ``cfg.MODEL.DEVICE``.  No weights are loaded here.  An unknown name is a ``KeyError``."""
import torch

from ...utils.registry import Registry

META_ARCH_REGISTRY = Registry("META_ARCH")


def build_model(cfg):
    arch_cls = META_ARCH_REGISTRY.get(cfg.MODEL.META_ARCHITECTURE)
    return arch_cls(cfg).to(torch.device(cfg.MODEL.DEVICE))
