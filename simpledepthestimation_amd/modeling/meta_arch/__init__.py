from .build import META_ARCH_REGISTRY, build_model  # noqa: F401
from .Supervised import SupDepthModel  # noqa: F401
from .MonoDepth2 import MonoDepth2Model  # noqa: F401
