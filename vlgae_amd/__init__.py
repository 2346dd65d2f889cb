"""vlgae_amd -- MI355X-native structured-DP hot path of VLGAE (LouChao98/VLGAE).

Sub-modules
    torch_struct   drop-in for the reference's `src.model.torch_struct` (DMV1o, DependencyCRF, ...)
    align          region x word bilinear alignment (`gather_logit_simple`) and the attention-fuse
    langfeat       `lang_feat_max_tree` / `lang_feat_word_only` as fused stages (encoders with SharedDropout masks, arc encoder)
    scorer         score construction feeding the DP (factorised-bilinear scores -> merged potentials)
    parser_ff      the parser's feed-forwards in front of it (head_ff / mid_ff / scorer projections)
    vis_encoder    the visual encoder's pairwise relation features
    feed           token-budget batch sampler and region-feature collate (host C++)
    dist           batch sharding + the single RCCL gradient all-reduce
    build          hipcc driver for the in-tree HIP extension

Everything computes through hand-written gfx950 kernels behind the C ABI of include/vlgae_amd.h.
"""
__version__ = "0.1.3"


def configure_autograd():
    """Run autograd's backward pass on the calling thread: `torch.autograd.set_multithreading_enabled(False)`.

    torch hands the backward of a GPU graph to a per-device engine thread; the hand-off is a condition-variable wake-up that
    costs 50-120 us per `backward()` / `autograd.grad()` call on these hosts -- more than the fused DP kernel takes (82 us).
    A one-process-per-GPU trainer has no use for that thread.  Call this once at start-up (INTEGRATION.md section 2 puts it
    next to the import alias); `vlgae_amd.torch_struct` warns once if it sees its backward running on another thread."""
    import torch
    torch.autograd.set_multithreading_enabled(False)
