"""Data side of the path: synthetic KITTI-shaped batches for benchmarks and tests; the KITTI reader and the preprocess
steps of detectron2/data (datasets/kitti_v2.py, preprocess/*.py)."""
