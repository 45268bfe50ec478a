from .build import PREPROCESS_REGISTRY, Preprocess, build_preprocess  # noqa: F401
from . import augmentation, formating, loading  # noqa: F401  (registers the steps)
