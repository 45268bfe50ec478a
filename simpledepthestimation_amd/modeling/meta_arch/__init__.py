"""Meta architectures of the hot path.  Importing the two model modules registers their classes in META_ARCH_REGISTRY."""
from . import MonoDepth2 as _mono, Supervised as _sup, build as _build

META_ARCH_REGISTRY, build_model = _build.META_ARCH_REGISTRY, _build.build_model
SupDepthModel, MonoDepth2Model = _sup.SupDepthModel, _mono.MonoDepth2Model
