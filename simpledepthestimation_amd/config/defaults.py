"""Default config values of the keys on the path (reference: detectron2/config/defaults.py).

COMPUTE_DTYPE is new: "fp32" reproduces the reference numerics (parity mode), "bf16" stores activations/weights of the
convolutions in bf16 with fp32 accumulation, statistics and losses (BASELINE.json config 2).
"""
from .config import CfgNode as CN

_C = CN()
_C.VERSION = 2
_C.MODEL = CN()
_C.MODEL.META_ARCHITECTURE = "SupDepthModel"
_C.MODEL.DEVICE = "cuda"
_C.MODEL.WEIGHTS = ""
_C.MODEL.MAX_DEPTH = 80
_C.MODEL.PIXEL_MEAN = [0.485, 0.456, 0.406]
_C.MODEL.PIXEL_STD = [0.229, 0.224, 0.225]
_C.MODEL.COMPUTE_DTYPE = "fp32"
_C.MODEL.DEPTH_NET = CN()
_C.MODEL.DEPTH_NET.NAME = "DepthResNet"
_C.MODEL.DEPTH_NET.ENCODER_NAME = "18"
_C.MODEL.DEPTH_NET.UPSAMPLE_DEPTH = False
_C.MODEL.DEPTH_NET.VERSION = "1A"          # PackNet01 only (packnet_1a.yaml)
_C.MODEL.POSE_NET = CN()
_C.MODEL.POSE_NET.NAME = "PoseNet"
_C.MODEL.POSE_NET.NUM_CONTEXTS = 2
_C.LOSS = CN()
_C.LOSS.SSIM_WEIGHT = 0.85
_C.LOSS.C1 = 1e-4
_C.LOSS.C2 = 9e-4
_C.LOSS.CLIP = 0.0
_C.LOSS.AUTOMASK = True
_C.LOSS.SMOOTHNESS_WEIGHT = 1e-3
_C.LOSS.PHOTOMETRIC_REDUCE = "min"
_C.LOSS.SUPERVISED_WEIGHT = 0.0
_C.LOSS.VARIANCE_FOCUS = 0.85
_C.LOSS.VAR_LOSS_WEIGHT = 0.0
_C.SOLVER = CN()
# --- values of detectron2/config/defaults.py:L101-131 for the keys the reference defines (merging one of its YAMLs here gives what merging it there gives) ---
_C.SOLVER.IMS_PER_BATCH = 16
_C.SOLVER.DEPTH_LR = 1e-3
_C.SOLVER.MAX_EPOCHS = 10
_C.SOLVER.CHECKPOINT_PERIOD = 1
_C.SOLVER.REFERENCE_WORLD_SIZE = 0
# --- keys the reference's defaults do not define (its project YAMLs add them through set_new_allowed, utils/setup.py:L18): neutral values ---
_C.SOLVER.POSE_LR = 1e-3
_C.SOLVER.DEPTH_END_LR = 1e-5
_C.SOLVER.LR_STEPS = ()
_C.SOLVER.GAMMA = 0.1
_C.SOLVER.AMP = False                      # fp16 storage + dynamic loss scaling (engine/train_loop.py:L294-341 AMPTrainer); needs MODEL.COMPUTE_DTYPE fp16
_C.TEST = CN()
_C.TEST.EVAL_PERIOD = 1
_C.TEST.GT_SCALE = False
_C.EVALUATORS = ("",)
_C.INPUT = CN()
_C.DATASETS = CN()
for _split in ("TRAIN", "TEST"):
    _C.DATASETS[_split] = CN()
    _C.DATASETS[_split].NAME = ""
    _C.DATASETS[_split].SPLIT = ""
    _C.DATASETS[_split].DATA_ROOT = ""
    _C.DATASETS[_split].IMG_WIDTH = 768
    _C.DATASETS[_split].IMG_HEIGHT = 384
    _C.DATASETS[_split].PREPROCESS = []
_C.DATALOADER = CN()
_C.DATALOADER.NUM_WORKERS = 6
_C.DATALOADER.SAMPLER_TRAIN = "DDPSampler"
_C.OUTPUT_DIR = "./output"
_C.SEED = -1
_C.CUDNN_BENCHMARK = False
_C.VIS_PERIOD = 0
_C.RUN_NAME = ""
_C.LOG_PERIOD = 20

# The hot-path keys of the two projects' Base.yaml (projects/MonoDepth2/configs/Base.yaml, projects/Supervised/configs/Base.yaml), as dict
# literals: /root/reference does not exist on the GPU box, so tests and bench.py build their configs from these instead of the YAML files
# (tests/test_config.py checks them against the files where the reference is present).
PROJECT_BASE = {
    "MonoDepth2": {
        "MODEL": {"META_ARCHITECTURE": "MonoDepth2Model", "MAX_DEPTH": 80},
        "LOSS": {"SSIM_WEIGHT": 0.85, "C1": 1e-4, "C2": 9e-4, "CLIP": 0.0, "AUTOMASK": True, "SMOOTHNESS_WEIGHT": 1e-3, "PHOTOMETRIC_REDUCE": "min",
                 "SUPERVISED_WEIGHT": 0.0, "VARIANCE_FOCUS": 0.85, "VAR_LOSS_WEIGHT": 0.0},
        "DATASETS": {"TEST": {"PREPROCESS": [{"NAME": "LoadImg"}, {"NAME": "LoadDepth", "KEEP_ORIG": True}, {"NAME": "ClipDepth", "MAX_DEPTH": 80},
                                             {"NAME": "Resize", "IMG_W": 640, "IMG_H": 192}, {"NAME": "ToTensor"}]}},
        "SOLVER": {"IMS_PER_BATCH": 16, "DEPTH_LR": 2e-4, "POSE_LR": 2e-4, "LR_STEPS": (15,), "GAMMA": 0.1, "MAX_EPOCHS": 20, "CHECKPOINT_PERIOD": 1},
        "EVALUATORS": ("kitti_evaluator", "kitti_evaluator_0_30", "kitti_evaluator_30_50", "kitti_evaluator_50_80"),
        "TEST": {"GT_SCALE": True},
        "LOG_PERIOD": 20,
    },
    "Supervised": {
        "MODEL": {"META_ARCHITECTURE": "SupDepthModel", "MAX_DEPTH": 80},
        "LOSS": {"VARIANCE_FOCUS": 0.85},
        "DATASETS": {"TEST": {"PREPROCESS": [{"NAME": "LoadImg"}, {"NAME": "LoadDepth", "KEEP_ORIG": True}, {"NAME": "ClipDepth", "MAX_DEPTH": 80},
                                             {"NAME": "KBCrop"}, {"NAME": "ToTensor"}]}},
        "SOLVER": {"IMS_PER_BATCH": 16, "DEPTH_LR": 1e-4, "DEPTH_END_LR": 1e-5, "MAX_EPOCHS": 50, "CHECKPOINT_PERIOD": 1},
        "EVALUATORS": ("kitti_evaluator", "kitti_evaluator_0_30", "kitti_evaluator_30_50", "kitti_evaluator_50_80"),
        "TEST": {"GT_SCALE": False},
        "LOG_PERIOD": 20,
    },
}
