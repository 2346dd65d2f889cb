"""vlgae_amd -- MI355X-native structured-DP hot path of VLGAE (LouChao98/VLGAE).

Sub-modules
    torch_struct   drop-in for the reference's `src.model.torch_struct` (DMV1o, DependencyCRF, ...)
    align          region x word bilinear alignment (`gather_logit_simple`) and the attention-fuse
    dist           batch sharding + the single RCCL gradient all-reduce
    build          hipcc driver for the in-tree HIP extension

Everything computes through hand-written gfx950 kernels behind the C ABI of include/vlgae_amd.h.
"""
__version__ = "0.1.0"
