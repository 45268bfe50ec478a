from .config import CfgNode, get_cfg  # noqa: F401
