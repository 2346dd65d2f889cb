"""CPU: the per-thread bodies of the HIP kernels (vlgae_amd/csrc/vlg_dp_core.h), executed by the host
phase emulator, against golden vectors from the reference and against the oracle.

What this proves without a GPU: index arithmetic, phase ordering, the ownership argument of the
outside pass (results must be bit-identical for every intra-phase thread order and thread count), and
that no phase touches memory outside the kernel's own chart layout (canary).
"""
import numpy as np
import pytest

from conftest import golden_files, golden_ids, load
from emu import emu

# lane counts must be powers of two (groups of G = 2^k lanes, like wavefronts)


# Expected counts are O(1) probabilities computed as exp(t - out) with fp32 charts: at |logZ| ~ 170 one ulp
# of a chart value is 1.5e-5, and the REFERENCE's own fp32 path deviates 1.7e-5 from its fp64 path on the
# hardest fixture (dmv_B4_L40_s1_full).  North-star bound: 1e-4.
MARG_TOL = 5e-5


def _tol(ref):
    return 2e-5 * np.maximum(1.0, np.abs(ref))


@pytest.mark.parametrize("path", [p for p in golden_files("dmv_") if "L80" not in p and "B8_L40" not in p],
                         ids=[i for i in golden_ids("dmv_") if "L80" not in i and "B8_L40" not in i])
def test_emu_dmv1o_golden(oracle_mod, path):
    g = load(path)
    md, ma = oracle_mod.dmv1o_merge(g["dec"], g["attach"], g["root"])
    lz, gd, ga = emu.dmv1o(md, ma, g["lengths"], 0, nt=16, order=0)
    assert np.all(np.abs(lz - g["logZ64"][:, 0]) <= _tol(g["logZ64"][:, 0]))
    assert np.abs(gd - g["grad_dec64"]).max() <= MARG_TOL
    assert np.abs(ga - g["grad_attach64"]).max() <= MARG_TOL
    # thread-order independence, bit for bit (no intra-phase races, no atomics)
    for order in (1, 5):
        lz2, gd2, ga2 = emu.dmv1o(md, ma, g["lengths"], 0, nt=16, order=order)
        assert np.array_equal(lz, lz2) and np.array_equal(gd, gd2) and np.array_equal(ga, ga2)
    # a different lane count changes the split of r over lanes (summation tree), not the mathematics
    lzw, gdw, gaw = emu.dmv1o(md, ma, g["lengths"], 0, nt=64, order=2)
    assert np.all(np.abs(lzw - g["logZ64"][:, 0]) <= _tol(g["logZ64"][:, 0]))
    assert np.abs(gdw - g["grad_dec64"]).max() <= MARG_TOL and np.abs(gaw - g["grad_attach64"]).max() <= MARG_TOL
    # inside-only kernel: same logZ bit for bit, and must not touch the outside-pass tape
    lz3, _, _ = emu.dmv1o(md, ma, g["lengths"], 0, grad=False, nt=16, order=2)
    assert np.array_equal(lz, lz3)
    # Max semiring: exact
    mz, mgd, mga = emu.dmv1o(md, ma, g["lengths"], 1, nt=16, order=3)
    assert np.allclose(mz, g["max"][:, 0], rtol=1e-6, atol=1e-6)
    assert np.array_equal(mgd, g["maxgrad_dec"]) and np.array_equal(mga, g["maxgrad_attach"])
    mz2, _, _ = emu.dmv1o(md, ma, g["lengths"], 1, grad=False, nt=8, order=1)
    assert np.array_equal(mz, mz2)   # max is exact: independent of the lane count
    # overlay placement (DmvLayout mode 1: value charts copied out, adjoint charts laid over them): same bits, both semirings
    for sr, want in ((0, (lz, gd, ga)), (1, (mz, mgd, mga))):
        got = emu.dmv1o(md, ma, g["lengths"], sr, nt=16, order=3 if sr else 0, mode=1)
        assert all(np.array_equal(a, b) for a, b in zip(got, want))
    # upstream gradient scaling
    _, wgd, wga = emu.dmv1o(md, ma, g["lengths"], 0, glogZ=g["wts"], nt=16)
    assert np.abs(wgd - g["wgrad_dec"]).max() <= 4e-5 and np.abs(wga - g["wgrad_attach"]).max() <= 4e-5
    assert emu.canary_trips() == 0


def test_emu_dmv1o_bf16_inputs(oracle_mod):
    g = load(golden_files("dmv_B4_L10_s0")[0])
    md, ma = oracle_mod.dmv1o_merge(g["dec"], g["attach"], g["root"])
    bits_d, bits_a = emu._bf16_bits(md), emu._bf16_bits(ma)
    md16 = (bits_d.astype(np.uint32) << 16).view(np.float32)
    ma16 = (bits_a.astype(np.uint32) << 16).view(np.float32)
    ref_lz, ref_gd, ref_ga = oracle_mod.dmv1o(md16, ma16, g["lengths"], "log", np.float64)
    lz, gd, ga = emu.dmv1o(md, ma, g["lengths"], 0, bf16=True)
    assert np.all(np.abs(lz - ref_lz[:, 0]) <= _tol(ref_lz[:, 0]))
    assert np.abs(gd - ref_gd).max() <= MARG_TOL and np.abs(ga - ref_ga).max() <= MARG_TOL


@pytest.mark.parametrize("path", [p for p in golden_files("deptree_") if "N81" not in p],
                         ids=[i for i in golden_ids("deptree_") if "N81" not in i])
def test_emu_deptree_golden(path):
    g = load(path)
    lz, ga = emu.deptree(g["arc"], g["lengths"], 0, nt=16, order=0)
    assert np.all(np.abs(lz - g["logZ64"]) <= _tol(g["logZ64"]))
    assert np.abs(ga - g["marginals64"]).max() <= MARG_TOL
    lz2, ga2 = emu.deptree(g["arc"], g["lengths"], 0, nt=16, order=1)
    assert np.array_equal(lz, lz2) and np.array_equal(ga, ga2)
    lz3, _ = emu.deptree(g["arc"], g["lengths"], 0, grad=False, nt=16, order=4)
    assert np.array_equal(lz, lz3)
    lzw, gaw = emu.deptree(g["arc"], g["lengths"], 0, nt=64, order=3)
    assert np.all(np.abs(lzw - g["logZ64"]) <= _tol(g["logZ64"])) and np.abs(gaw - g["marginals64"]).max() <= MARG_TOL
    mz, mga = emu.deptree(g["arc"], g["lengths"], 1, nt=16, order=2)
    assert np.allclose(mz, g["max"], rtol=1e-6, atol=1e-6) and np.array_equal(mga, g["argmax"])
    mz2, _ = emu.deptree(g["arc"], g["lengths"], 1, grad=False, nt=4)
    assert np.array_equal(mz, mz2)
    _, wg = emu.deptree(g["arc"], g["lengths"], 0, glogZ=g["wts"])
    assert np.abs(wg - g["wgrad"]).max() <= 4e-5
    if np.all(g["lengths"] == g["arc"].shape[1] - 1):
        lz4, _ = emu.deptree(g["arc"], None, 0, grad=False)
        assert np.array_equal(lz, lz4)
    assert emu.canary_trips() == 0


def test_emu_ties_take_first_argmax(oracle_mod):
    """All-equal potentials: every tree ties.  torch.max's backward goes to the FIRST maximal index
    (semirings.py:199-200); the kernels' back-pointers must reproduce the same single tree."""
    B, N = 2, 7
    md = np.zeros((B, N, 2, 2, 2), np.float32)
    ma = np.zeros((B, N, N, 2), np.float32)
    ma[:, :, 0, :] = -1e12      # the root is nobody's child
    md[:, 0, 0] = -1e12         # and has no LEFT decisions (merge's fill)
    ma[:, 0, :, 0] = -1e12      # root attaches with valence NOCHILD only
    lengths = np.array([6, 4])
    ref = oracle_mod.dmv1o(md, ma, lengths, "max", np.float32)
    mz, mgd, mga = emu.dmv1o(md, ma, lengths, 1, nt=8, order=1)
    assert np.array_equal(mz, ref[0][:, 0]) and np.array_equal(mgd, ref[1]) and np.array_equal(mga, ref[2])
    assert np.array_equal(mga.sum((1, 2, 3)), lengths.astype(np.float32))     # still a tree


def test_emu_decode_heads(oracle_mod):
    """Decode mode: heads written by the kernel body == the callers' nonzero()+scatter on the reference's argmax."""
    g = load(golden_files("dmv_B4_L10_s0")[0])
    md, ma = oracle_mod.dmv1o_merge(g["dec"], g["attach"], g["root"])
    best, heads = emu.dmv1o_decode(md, ma, g["lengths"], nt=16, order=1)
    assert np.array_equal(heads, g["predicted"]) and np.allclose(best, g["max"][:, 0], rtol=1e-6, atol=1e-6)
    # MBR: DependencyCRF over the reference's arc marginals
    best2, heads2 = emu.deptree_decode(g["arc_marginal"], g["lengths"], nt=16, order=2)
    B, N = heads2.shape
    ref = np.zeros((B, N), np.int64)
    b, h, c = np.nonzero(g["mbr_argmax"])
    ref[b, c] = h
    assert np.array_equal(heads2, ref) and np.allclose(best2, g["mbr_max"], rtol=1e-6, atol=1e-6)
    assert emu.canary_trips() == 0


@pytest.mark.parametrize("path", [p for p in golden_files("rules_") if "L40" not in p],
                         ids=[i for i in golden_ids("rules_") if "L40" not in i])
def test_emu_rules_golden(oracle_mod, path):
    """Rule-table input path (RuleIO): gather / direction select / function mask / merge folded into the load
    stage, rule-space expected counts out -- against the reference's own ops (ldndmv.py:185-209 + DMV1o)."""
    g = load(path)
    hm = g["head_mask"] if g["head_mask"].any() else None
    lz, gr, gd, groot, _ = emu.dmv1o_rules(g["attach_rule"], g["dec"], g["root_rule"], g["token"], g["lengths"], hm,
                                           nt=16, order=0)
    assert np.all(np.abs(lz - g["logZ64"][:, 0]) <= _tol(g["logZ64"][:, 0]))
    assert np.abs(gr - g["grad_rule64"]).max() <= MARG_TOL and np.abs(gd - g["grad_dec64"]).max() <= MARG_TOL
    got_root = groot if int(g["root_per_sentence"]) else groot.sum(0, keepdims=True)
    assert np.abs(got_root - g["grad_root64"]).max() <= 4 * MARG_TOL
    lz2, gr2, gd2, groot2, _ = emu.dmv1o_rules(g["attach_rule"], g["dec"], g["root_rule"], g["token"], g["lengths"], hm,
                                               nt=16, order=3)
    assert np.array_equal(lz, lz2) and np.array_equal(gd, gd2)
    assert np.abs(gr - gr2).max() <= 1e-6        # scatter-add over repeated tokens: summation order may differ
    lz3, *_ = emu.dmv1o_rules(g["attach_rule"], g["dec"], g["root_rule"], g["token"], g["lengths"], hm, grad=False)
    assert np.array_equal(lz, lz3)
    mz, _, _, _, heads = emu.dmv1o_rules(g["attach_rule"], g["dec"], g["root_rule"], g["token"], g["lengths"], hm,
                                         semiring=1, grad=False, heads=True)
    assert np.allclose(mz, g["max"][:, 0], rtol=1e-6, atol=1e-5)
    # repeated tokens make equal-score trees common in rule space: compare the Viterbi trees by VALUE
    for b, ln in enumerate(g["lengths"]):
        assert oracle_mod.is_projective_tree(heads[b], int(ln)) and np.all(heads[b, ln + 1:] == 0)
        sc = oracle_mod.dmv1o_tree_score(g["merged_dec"][b], g["merged_attach"][b], heads[b], int(ln))
        assert abs(sc - g["max"][b, 0]) <= 1e-4 * max(1.0, abs(g["max"][b, 0]))
    assert emu.canary_trips() == 0
