"""Minimal yacs-style config: enough to load the reference's projects/*/configs/*.yaml unchanged.

Mirrors the behaviour the path relies on (detectron2/config/config.py:L12-84, utils/setup.py:L17-20): attribute access,
``_BASE_`` inheritance relative to the including file, YAML anchors, ``merge_from_list(['KEY.SUB', 'value', ...])``,
``set_new_allowed`` and ``freeze``.  Tuples written as ``(15,)`` strings in YAML are evaluated like yacs does.
"""
import ast
import copy
import os

import yaml

BASE_KEY = "_BASE_"


class CfgNode(dict):
    def __init__(self, init=None, new_allowed=True):
        super().__init__()
        object.__setattr__(self, "_frozen", False)
        object.__setattr__(self, "_new_allowed", new_allowed)
        for k, v in (init or {}).items():
            self[k] = CfgNode(v, new_allowed) if isinstance(v, dict) and not isinstance(v, CfgNode) else v

    def __getattr__(self, name):
        try:
            return self[name]
        except KeyError:
            raise AttributeError(name)

    def __setattr__(self, name, value):
        if self._frozen:
            raise AttributeError(f"Attempted to set {name} to {value}, but CfgNode is immutable")
        self[name] = value

    def set_new_allowed(self, flag):
        object.__setattr__(self, "_new_allowed", flag)
        for v in self.values():
            if isinstance(v, CfgNode):
                v.set_new_allowed(flag)

    def freeze(self):
        object.__setattr__(self, "_frozen", True)
        for v in self.values():
            if isinstance(v, CfgNode):
                v.freeze()

    def defrost(self):
        object.__setattr__(self, "_frozen", False)
        for v in self.values():
            if isinstance(v, CfgNode):
                v.defrost()

    def is_frozen(self):
        return self._frozen

    def clone(self):
        return copy.deepcopy(self)

    def __deepcopy__(self, memo):
        out = CfgNode(new_allowed=self._new_allowed)
        for k, v in self.items():
            dict.__setitem__(out, k, copy.deepcopy(v, memo))
        return out

    @staticmethod
    def _decode(v):
        if isinstance(v, str):
            try:
                return ast.literal_eval(v)
            except (ValueError, SyntaxError):
                return v
        return v

    @classmethod
    def load_yaml_with_base(cls, filename):
        with open(filename, "r") as f:
            cfg = yaml.safe_load(f) or {}
        if BASE_KEY in cfg:
            base = cfg.pop(BASE_KEY)
            if base.startswith("~"):
                base = os.path.expanduser(base)
            if not base.startswith("/"):
                base = os.path.join(os.path.dirname(filename), base)
            merged = cls.load_yaml_with_base(base)
            _merge_dict(cfg, merged)
            return merged
        return cfg

    def merge_from_file(self, filename):
        self.merge_from_other_cfg(self.load_yaml_with_base(filename))

    def merge_from_other_cfg(self, other):
        _merge_into(other, self, [])

    def merge_from_list(self, opts):
        if len(opts) % 2:
            raise AssertionError(f"Override list has odd length: {opts}; it must be a list of pairs")
        for full_key, v in zip(opts[0::2], opts[1::2]):
            d = self
            keys = full_key.split(".")
            for sub in keys[:-1]:
                if sub not in d:
                    if not self._new_allowed:
                        raise KeyError(f"Non-existent config key: {full_key}")
                    d[sub] = CfgNode(new_allowed=self._new_allowed)
                d = d[sub]
            if keys[-1] not in d and not self._new_allowed:
                raise KeyError(f"Non-existent config key: {full_key}")
            d[keys[-1]] = self._decode(v)

    def dump(self):
        def plain(n):
            return {k: plain(v) if isinstance(v, CfgNode) else (list(v) if isinstance(v, tuple) else v) for k, v in n.items()}
        return yaml.safe_dump(plain(self))


def _merge_dict(a, b):
    """merge plain dict a into b (a wins)."""
    for k, v in a.items():
        if isinstance(v, dict) and isinstance(b.get(k), dict):
            _merge_dict(v, b[k])
        else:
            b[k] = v


def _merge_into(a, b, key_list):
    for k, v in a.items():
        full = ".".join(key_list + [k])
        if isinstance(v, dict):
            if k in b and isinstance(b[k], CfgNode):
                _merge_into(v, b[k], key_list + [k])
                continue
            v = CfgNode(v, b._new_allowed)
        elif k not in b and not b._new_allowed:
            raise KeyError(f"Non-existent config key: {full}")
        else:
            v = CfgNode._decode(v)
        b[k] = v


def get_cfg():
    """Defaults for the keys the hot path reads (reference: detectron2/config/defaults.py:L25-40,L138-157 and SURVEY.md 8b)."""
    from .defaults import _C
    return _C.clone()


def get_project_cfg(project):
    """get_cfg() + the hot-path keys of projects/<project>/configs/Base.yaml ("MonoDepth2" or "Supervised"), from the embedded literals."""
    from .defaults import PROJECT_BASE
    cfg = get_cfg()
    cfg.merge_from_other_cfg(copy.deepcopy(PROJECT_BASE[project]))
    return cfg
