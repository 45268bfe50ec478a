"""Data side of the path: the KITTI reader and preprocess steps of detectron2/data (datasets/kitti_v2.py,
preprocess/*.py, build.py), synthetic KITTI-shaped batches for benchmarks and tests, and a pinned-memory device prefetcher."""
from .build import (DATASET_REGISTRY, DatasetBase, DevicePrefetcher, InferenceSampler, build_batch_data_loader,  # noqa: F401
                    build_detection_test_loader, build_detection_train_loader)
from . import datasets  # noqa: F401  (registers KittiDepthV2)
