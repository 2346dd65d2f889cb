"""CPU oracle for the VLGAE structured-DP hot path -- TEST INFRASTRUCTURE, not product code.

Only `tests/`, `__graft_entry__.smoke()` and `bench.py`'s `cpu_baseline` leg may import this
package.  The product (`vlgae_amd`) never does, and raises if its HIP library is missing.
"""
from .cpu_oracle import (  # noqa: F401
    NEGINF, arc_encoder, arc_encoder_backward, attn_fuse, attn_fuse_backward, bilinear_align, bilinear_align_backward, box_rel, build, deptree, dmv1o, dmv1o_merge, dmv1o_rules, dmv1o_tree_score, enumerate_dmv1o, is_projective_tree,
    enumerate_deptree, feed_batches, feed_kmeans, gather_logit_reduced, grounding_decode, grounding_decode_lists, grounding_loss, grounding_prior, lang_feat, lang_feat_marginal, max_threads, mlp, ndmv_potentials, set_threads,
)
