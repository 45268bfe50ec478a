"""Checkpoints in the reference's file format (see checkpoint.py)."""
from . import checkpoint as _impl

DetectionCheckpointer = _impl.DetectionCheckpointer
PeriodicCheckpointer = _impl.PeriodicCheckpointer
