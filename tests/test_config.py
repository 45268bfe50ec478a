"""Config surface (detectron2/config/defaults.py, utils/setup.py:L17-20): defaults equal the reference's for the keys it defines, the
embedded project bases equal projects/*/configs/Base.yaml, and every YAML of the two projects on the path merges unchanged.

The YAML files exist only where /root/reference does (the build container): those tests skip elsewhere."""
import glob
import os

import pytest

from simpledepthestimation_amd.config import get_cfg, get_project_cfg
from simpledepthestimation_amd.config.defaults import PROJECT_BASE

REF = "/root/reference"
needs_ref = pytest.mark.skipif(not os.path.isdir(os.path.join(REF, "projects")), reason="reference checkout not present (GPU box)")


def test_defaults_are_the_references():
    c = get_cfg()
    assert c.TEST.EVAL_PERIOD == 1 and c.DATASETS.TEST.PREPROCESS == [] and c.DATASETS.TRAIN.PREPROCESS == []
    assert c.SOLVER.DEPTH_LR == 1e-3 and c.SOLVER.MAX_EPOCHS == 10 and c.SOLVER.IMS_PER_BATCH == 16 and c.SOLVER.CHECKPOINT_PERIOD == 1
    assert tuple(c.EVALUATORS) == ("",) and c.DATALOADER.NUM_WORKERS == 6 and c.LOG_PERIOD == 20 and c.MODEL.MAX_DEPTH == 80
    assert c.MODEL.PIXEL_MEAN == [0.485, 0.456, 0.406] and c.MODEL.PIXEL_STD == [0.229, 0.224, 0.225]


def _flat(d, prefix=""):
    out = {}
    for k, v in d.items():
        if isinstance(v, dict):
            out.update(_flat(v, prefix + k + "."))
        else:
            out[prefix + k] = tuple(v) if isinstance(v, (list, tuple)) and not (v and isinstance(v[0], dict)) else v
    return out


@needs_ref
@pytest.mark.parametrize("project", ["MonoDepth2", "Supervised"])
def test_embedded_project_base_equals_the_yaml(project):
    """Every key of PROJECT_BASE[project] has the value projects/<project>/configs/Base.yaml gives it."""
    cfg = get_cfg()
    cfg.merge_from_file(os.path.join(REF, "projects", project, "configs", "Base.yaml"))
    want = _flat(cfg)
    for k, v in _flat(PROJECT_BASE[project]).items():
        if k == "DATASETS.TEST.PREPROCESS":
            got = [dict(s) for s in cfg.DATASETS.TEST.PREPROCESS]
            assert got == [dict(s) for s in v], k
        else:
            assert want[k] == v, (k, want[k], v)
    # and the embedded form resolves to the same hot-path values as the file
    emb = _flat(get_project_cfg(project))
    for k in ("SOLVER.DEPTH_LR", "SOLVER.MAX_EPOCHS", "TEST.GT_SCALE", "TEST.EVAL_PERIOD", "EVALUATORS", "LOG_PERIOD", "MODEL.META_ARCHITECTURE"):
        assert emb[k] == want[k], k


@needs_ref
@pytest.mark.parametrize("path", sorted(glob.glob(os.path.join(REF, "projects", "Supervised", "configs", "*.yaml")) +
                                        glob.glob(os.path.join(REF, "projects", "MonoDepth2", "configs", "*.yaml"))),
                         ids=lambda p: "/".join(p.split("/")[-3::2]))
def test_reference_yaml_merges_unchanged(path):
    """the reference's YAMLs load as they are (``_BASE_`` inheritance, anchors, tuple strings, new keys) and CLI-style overrides apply."""
    cfg = get_cfg()
    cfg.set_new_allowed(True)                       # utils/setup.py:L18
    cfg.merge_from_file(path)
    cfg.merge_from_list(["SOLVER.IMS_PER_BATCH", "24", "MODEL.DEPTH_NET.ENCODER_NAME", "18", "OUTPUT_DIR", "/tmp/x"])
    assert cfg.SOLVER.IMS_PER_BATCH == 24 and cfg.MODEL.DEPTH_NET.ENCODER_NAME == 18 or cfg.MODEL.DEPTH_NET.ENCODER_NAME == "18"
    assert cfg.MODEL.META_ARCHITECTURE in ("SupDepthModel", "MonoDepth2Model")
    assert cfg.MODEL.MAX_DEPTH == 80 and cfg.TEST.EVAL_PERIOD == 1           # EVAL_PERIOD is never set by a YAML: the reference's default must survive
    names = [s["NAME"] for s in cfg.DATASETS.TEST.PREPROCESS]
    assert names[0] == "LoadImg" and names[-1] == "ToTensor" and ("Resize" in names or "KBCrop" in names or "CropTopTo" in names)
    assert isinstance(cfg.EVALUATORS, tuple) and cfg.EVALUATORS[0] == "kitti_evaluator"
    if "MonoDepth2" in path:
        assert isinstance(cfg.SOLVER.LR_STEPS, tuple) and cfg.LOSS.SSIM_WEIGHT == 0.85 and cfg.SOLVER.GAMMA == 0.1
        if "waymo" not in path:
            assert cfg.SOLVER.LR_STEPS == (15,) and cfg.SOLVER.DEPTH_LR == 2e-4 and cfg.TEST.GT_SCALE is True
        if "Base" not in os.path.basename(path):
            assert cfg.MODEL.POSE_NET.NUM_CONTEXTS == 2 and cfg.MODEL.POSE_NET.NAME == "PoseNet"
    cfg.freeze()
    with pytest.raises(AttributeError):
        cfg.OUTPUT_DIR = "y"
