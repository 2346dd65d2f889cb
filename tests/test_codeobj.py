"""CPU: resource hygiene of the shipped kernels, read from the gfx950 code-object notes of the built objects
(tools/codeobj.py): the kernels on the benchmarked paths use no private scratch and spill no vector registers.  A kernel
that starts to spill shows up here at build time instead of as an unexplained slowdown on the GPU box."""
import os
import sys

import pytest

from conftest import ROOT

# kernels on the hot paths bench.py reports (substring of the demangled name)
HOT = ("dmv1o_kernel<", "dmv1o_rules_kernel<", "deptree_kernel<", "align_max_kernel<", "align_mfma_kernel<", "attn_fuse_mfma_kernel<",
       "attn_fuse_bwd_words_kernel<", "attn_fuse_bwd_regions_kernel<", "tri_kernel<false", "tri_dw2_kernel", "ground_bwd_dense_kernel<",
       "ground_ce_tile_kernel<", "gemm_tn_kernel", "scorer_fwd_kernel<", "scorer_bwd_kernel<", "langfeat_", "box_rel_", "merge_kernel<",
       "grounding_decode_kernel", "align_bwd_split_kernel<", "align_full_kernel", "tri2_kernel", "align_argmax_kernel<",
       "ground_bwd_ws_kernel<", "ground_ce_tile2_kernel", "align_prior_diag_kernel", "attn_fuse_split_kernel<", "attn_bwd_sweep_kernel<",
       "attn_bwd_combine_kernel<", "attn_fuse_combine_kernel<", "attn_bwd_regions_bf16_kernel<", "ff_gemm_act_kernel", "ff_gemm_act2_kernel",
       "gemm_tn_group_kernel")
# known exceptions, each with its reason (fallback paths the benchmarked configurations do not take, or work in progress)
ALLOWED = {
    "tri_dw_kernel<": "fallback of tri_dw2_kernel (fp32 features / widths other than 128): 128 accumulators + operand ring",
    "ground_bwd_kernel<": "sparse row-update fallback of ground_bwd_dense_kernel (fp32 features, very wide factor layouts)",
    "align_bwd_split_f32_kernel<": "a9 backward, fp32 features (both bf16 halves of the feature tile resident): 2 registers over its 256 cap",
    "align_bwd_split2_kernel<": "a9 backward, caption side with two images per step: 3 registers over its cap",
}


@pytest.fixture(scope="module")
def kernels():
    from vlgae_amd.build import build_library
    build_library()
    sys.path.insert(0, os.path.join(ROOT, "tools"))
    import codeobj
    ks = codeobj.all_kernels()
    assert len(ks) > 200, len(ks)
    return ks


def test_hot_kernels_have_no_scratch_and_no_vgpr_spills(kernels):
    seen = {h: 0 for h in HOT}
    bad = []
    for k in kernels:
        name = k.get("demangled", k.get("name", ""))
        for h in HOT:
            if h in name:
                seen[h] += 1
                if k.get("private_segment_fixed_size", 0) or k.get("vgpr_spill_count", 0):
                    bad.append((name[:100], k.get("private_segment_fixed_size"), k.get("vgpr_spill_count")))
    assert not bad, bad
    assert all(n > 0 for n in seen.values()), {h: n for h, n in seen.items() if n == 0}   # a renamed kernel must not drop out silently


def test_every_other_kernel_with_scratch_is_a_documented_exception(kernels):
    for k in kernels:
        name = k.get("demangled", k.get("name", ""))
        if k.get("private_segment_fixed_size", 0) or k.get("vgpr_spill_count", 0):
            assert any(a in name for a in ALLOWED), (name[:120], k.get("private_segment_fixed_size"), k.get("vgpr_spill_count"))


def test_register_budgets_match_the_launch_shapes(kernels):
    """A wavefront of a B-thread workgroup can hold 512 / ceil(B / 256) registers (arch + accumulation VGPRs): the compiler
    honours it, but a launch bound that silently halves occupancy is worth a look -- the headline DP kernel must keep
    two 512-thread workgroups per CU (<= 128 registers)."""
    for k in kernels:
        name = k.get("demangled", "")
        total = k.get("vgpr_count", 0)
        assert total <= 512, (name[:100], total)
        if "dmv1o_kernel<0, 0, true, vlg::BF16In>" in name:
            assert total <= 128, total


def test_round4_second_half_kernels_are_on_the_hot_list(kernels):
    """The kernels added after the list above was written: both DPs of lang_feat_max_tree as one launch, the element-wise passes of the
    parser's feed-forwards.  No scratch, no VGPR spills; the pair kernel within the 128 registers that let a sentence's two workgroups share a CU."""
    names = ("dmv1o_pair_kernel<", "ff_mlp_act_kernel<", "ff_act_kernel<", "ff_act_bwd_kernel<", "ff_mlp_act_bwd_kernel<", "small_gemm_kernel<")
    found = {}
    for k in kernels:
        dem = k.get("demangled", k.get("name", ""))
        for name in names:
            if name in dem:
                found[name] = found.get(name, 0) + 1
                assert not k.get("private_segment_fixed_size", 0) and not k.get("vgpr_spill_count", 0), dem
                if name == "dmv1o_pair_kernel<":
                    assert k.get("vgpr_count", 999) <= 128, (dem, k.get("vgpr_count"))
    assert len(found) == len(names), found
