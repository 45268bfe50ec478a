"""GPU evaluation of depth predictions: evaluator registry + the KITTI metric evaluators (importing this package registers them)."""
from . import depth_evaluation as _kitti
from . import evaluator as _base

EVALUATOR_REGISTRY = _base.EVALUATOR_REGISTRY
DatasetEvaluator, DatasetEvaluators = _base.DatasetEvaluator, _base.DatasetEvaluators
build_evaluator, inference_context, inference_on_dataset = _base.build_evaluator, _base.inference_context, _base.inference_on_dataset
kitti_evaluator = _kitti.kitti_evaluator
kitti_evaluator_0_30, kitti_evaluator_30_50, kitti_evaluator_50_80 = _kitti.kitti_evaluator_0_30, _kitti.kitti_evaluator_30_50, _kitti.kitti_evaluator_50_80
kitti_depth_saver, write_depth = _kitti.kitti_depth_saver, _kitti.write_depth
