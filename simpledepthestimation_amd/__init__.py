"""simpledepthestimation_amd -- MI355X (gfx950) native depth-training hot path.

Host-side mirror of the reference's plugin surface (build_model / registries / model(batch)->dict,
zzzxxxttt/SimpleDepthEstimation detectron2/modeling/meta_arch/build.py:L15-23) over a C-ABI HIP
library (include/sde_hip.h, built in-tree as simpledepthestimation_amd/libsde_hip.so).
All arithmetic on the path runs in that library; there is no CPU fallback.
"""
__version__ = "0.1.0"
