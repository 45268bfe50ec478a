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
_C.SOLVER.IMS_PER_BATCH = 16
_C.SOLVER.DEPTH_LR = 2e-4
_C.SOLVER.POSE_LR = 2e-4
_C.SOLVER.DEPTH_END_LR = 1e-5
_C.SOLVER.LR_STEPS = (15,)
_C.SOLVER.GAMMA = 0.1
_C.SOLVER.MAX_EPOCHS = 20
_C.SOLVER.CHECKPOINT_PERIOD = 1
_C.TEST = CN()
_C.TEST.EVAL_PERIOD = 0
_C.TEST.GT_SCALE = False
# evaluators run by inference_on_dataset (projects/*/configs/Base.yaml: EVALUATORS); DATASETS.TEST.PREPROCESS names the test-time preprocess
# steps whose backward() chain the evaluators fold into their index maps (MonoDepth2: Resize; Supervised: KBCrop)
_C.EVALUATORS = ("kitti_evaluator", "kitti_evaluator_0_30", "kitti_evaluator_30_50", "kitti_evaluator_50_80")
_C.DATASETS = CN()
_C.DATASETS.TEST = CN()
_C.DATASETS.TEST.PREPROCESS = [{"NAME": "Resize"}]
_C.DATALOADER = CN()
_C.DATALOADER.NUM_WORKERS = 4
_C.OUTPUT_DIR = "./output"
_C.SEED = -1
_C.RUN_NAME = ""
_C.LOG_PERIOD = 20
