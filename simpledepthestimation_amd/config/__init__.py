from .config import CfgNode, get_cfg, get_project_cfg  # noqa: F401
