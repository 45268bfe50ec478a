"""Preprocess registry (reference: detectron2/data/preprocess/build.py): ``PREPROCESS_REGISTRY``, ``Preprocess`` with forward / backward,
``build_preprocess(cfg)`` keyed by cfg.NAME.  Step configs are the YAML dicts of DATASETS.*.PREPROCESS (attribute and item access both work)."""
from ...utils.registry import Registry

PREPROCESS_REGISTRY = Registry("PREPROCESS")


class StepCfg(dict):
    """A preprocess step's config: dict with attribute access (the reference wraps it in an EasyDict)."""

    def __getattr__(self, k):
        try:
            return self[k]
        except KeyError:
            raise AttributeError(k)


class Preprocess:
    def __init__(self, cfg):
        self.cfg = cfg

    def forward(self, data_dict):
        return data_dict

    def backward(self, data_dict):
        return data_dict


def build_preprocess(cfg):
    cfg = StepCfg(cfg)
    preprocess = PREPROCESS_REGISTRY.get(cfg.NAME)(cfg)
    assert isinstance(preprocess, Preprocess)
    return preprocess
