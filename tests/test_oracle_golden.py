"""CPU: pin the oracle (oracle/vlg_oracle.c) on golden vectors produced by the reference itself
(tests/golden/make_golden.py) and on brute-force enumeration.  The reference has no tests or fixtures of
its own for this path; these vectors are what "parity" means in this repo.
"""
import numpy as np
import pytest

from conftest import golden_files, golden_ids, load


@pytest.mark.parametrize("path", golden_files("dmv_"), ids=golden_ids("dmv_"))
def test_oracle_dmv1o_matches_reference(oracle_mod, path):
    g = load(path)
    md, ma = oracle_mod.dmv1o_merge(g["dec"], g["attach"], g["root"])
    if "merged_dec" in g:
        assert np.array_equal(md, g["merged_dec"]) and np.array_equal(ma, g["merged_attach"])
    # fp64 oracle == reference run in fp64, to rounding
    lz, gd, ga = oracle_mod.dmv1o(md, ma, g["lengths"], "log", np.float64)
    assert np.abs(lz - g["logZ64"]).max() <= 1e-11
    assert np.abs(gd - g["grad_dec64"]).max() <= 1e-11 and np.abs(ga - g["grad_attach64"]).max() <= 1e-11
    mz, mgd, mga = oracle_mod.dmv1o(md, ma, g["lengths"], "max", np.float64)
    assert np.abs(mz - g["max64"]).max() <= 1e-11
    assert np.array_equal(mgd, g["maxgrad_dec64"]) and np.array_equal(mga, g["maxgrad_attach64"])
    # fp32 oracle vs reference fp32: both are fp32 evaluations of the same recurrences
    lz32, gd32, ga32 = oracle_mod.dmv1o(md, ma, g["lengths"], "log", np.float32)
    assert np.all(np.abs(lz32 - g["logZ"]) <= 4e-5 * np.maximum(1, np.abs(g["logZ"])))
    assert np.abs(gd32 - g["grad_dec"]).max() <= 5e-5 and np.abs(ga32 - g["grad_attach"]).max() <= 5e-5
    mz32, mgd32, mga32 = oracle_mod.dmv1o(md, ma, g["lengths"], "max", np.float32)
    assert np.allclose(mz32, g["max"], rtol=1e-6, atol=1e-6)
    assert np.array_equal(mgd32, g["maxgrad_dec"]) and np.array_equal(mga32, g["maxgrad_attach"])
    assert np.array_equal(mga32, g["argmax"])                       # .argmax == Max-semiring attach gradient
    assert np.abs(ga32 - g["marginals"]).max() <= 5e-5              # .marginals == Log-semiring attach gradient
    # upstream-gradient scaling
    _, wgd, wga = oracle_mod.dmv1o(md, ma, g["lengths"], "log", np.float32, glogZ=g["wts"])
    assert np.abs(wgd - g["wgrad_dec"]).max() <= 1e-4 and np.abs(wga - g["wgrad_attach"]).max() <= 1e-4
    # call-site contract (joint.py:256-258): predicted heads from the best tree
    B, N = md.shape[:2]
    pred = np.zeros((B, N), dtype=np.int64)
    b, h, c = np.nonzero(mga32.sum(-1))
    pred[b, c] = h
    assert np.array_equal(pred, g["predicted"])
    # MBR chain (ldndmv.py:294-299)
    mbr_max, mbr_arg = oracle_mod.deptree(g["arc_marginal"], g["lengths"], "max", np.float32)
    assert np.array_equal(mbr_arg, g["mbr_argmax"]) and np.allclose(mbr_max, g["mbr_max"], rtol=1e-6, atol=1e-6)


@pytest.mark.parametrize("path", golden_files("deptree_"), ids=golden_ids("deptree_"))
def test_oracle_deptree_matches_reference(oracle_mod, path):
    g = load(path)
    lz, ga = oracle_mod.deptree(g["arc"], g["lengths"], "log", np.float64)
    assert np.abs(lz - g["logZ64"]).max() <= 1e-11 and np.abs(ga - g["marginals64"]).max() <= 1e-11
    mz, mga = oracle_mod.deptree(g["arc"], g["lengths"], "max", np.float64)
    assert np.abs(mz - g["max64"]).max() <= 1e-11 and np.array_equal(mga, g["argmax64"])
    lz32, ga32 = oracle_mod.deptree(g["arc"], g["lengths"], "log", np.float32)
    assert np.all(np.abs(lz32 - g["logZ"]) <= 4e-5 * np.maximum(1, np.abs(g["logZ"])))
    assert np.abs(ga32 - g["marginals"]).max() <= 5e-5
    _, wg = oracle_mod.deptree(g["arc"], g["lengths"], "log", np.float32, glogZ=g["wts"])
    assert np.abs(wg - g["wgrad"]).max() <= 1e-4
    if "enum_logZ" in g:
        # the reference's own brute-force enumerator (deptree.py:213-228) and ours agree with the DP
        assert np.abs(lz - g["enum_logZ"]).max() <= 1e-10 and np.abs(mz - g["enum_max"]).max() <= 1e-10
        for b in range(len(lz)):
            e_lz, e_mx = oracle_mod.enumerate_deptree(g["arc"][b].astype(np.float64), int(g["lengths"][b]))
            assert abs(e_lz - lz[b]) <= 1e-10 and abs(e_mx - mz[b]) <= 1e-10


def test_oracle_dmv1o_bruteforce(oracle_mod):
    """The reference has no enumerator for DMV1o; ours scores every projective single-root tree with the
    valence rules read off dmv.py:36-62 and must agree with the DP (ragged lengths included)."""
    g = load(golden_files("dmv_B5_L7_s2")[0])
    md, ma = oracle_mod.dmv1o_merge(g["dec"], g["attach"], g["root"])
    lz, _, _ = oracle_mod.dmv1o(md, ma, g["lengths"], "log", np.float64, grad=False)
    mz, _, _ = oracle_mod.dmv1o(md, ma, g["lengths"], "max", np.float64, grad=False)
    for b, ln in enumerate(g["lengths"]):
        e_lz, e_mx = oracle_mod.enumerate_dmv1o(md[b].astype(np.float64), ma[b].astype(np.float64), int(ln))
        assert abs(e_lz - lz[b, 0]) <= 1e-10 and abs(e_mx - mz[b, 0]) <= 1e-10


def test_oracle_identities(oracle_mod):
    """SURVEY.md section 4 (i)-(v) on the oracle, B=16 L=40."""
    rng = np.random.default_rng(3)
    B, L = 16, 40
    dec = np.log(rng.dirichlet(np.ones(2), size=(B, L, 2, 2))).astype(np.float32)
    attach = rng.standard_normal((B, L, L, 2)).astype(np.float32)
    root = rng.standard_normal((B, L)).astype(np.float32)
    lengths = rng.integers(1, L + 1, size=B)
    lengths[0] = L
    md, ma = oracle_mod.dmv1o_merge(dec, attach, root)
    _, gd, ga = oracle_mod.dmv1o(md, ma, lengths, "log", np.float64)
    assert np.allclose(ga.sum((1, 2, 3)), lengths, atol=1e-9)                 # (ii)
    assert np.allclose(gd.sum((1, 2, 3, 4)), 3 * lengths + 1, atol=1e-9)      # (iii)
    for b, ln in enumerate(lengths):                                          # (v) exact zeros on padding
        assert np.all(ga[b, ln + 1:] == 0) and np.all(ga[b, :, ln + 1:] == 0) and np.all(gd[b, ln + 1:] == 0)
    arc = rng.standard_normal((B, L + 1, L + 1)).astype(np.float32)
    lz_crf, m = oracle_mod.deptree(arc, lengths, "log", np.float64)
    for b, ln in enumerate(lengths):                                          # (iv)
        assert np.allclose(m[b].sum(0)[1:ln + 1], 1.0, atol=1e-9) and np.all(m[b].sum(0)[ln + 1:] == 0)
    z = np.zeros((B, L + 1, 2, 2, 2), np.float32)                             # (i) DMV1o degenerates to the CRF
    lz_dmv, _, _ = oracle_mod.dmv1o(z, np.repeat(arc[..., None], 2, -1), lengths, "log", np.float64, grad=False)
    assert np.allclose(lz_dmv[:, 0], lz_crf, atol=1e-9)


@pytest.mark.parametrize("path", golden_files("align_"), ids=golden_ids("align_"))
def test_oracle_align_matches_reference(oracle_mod, path):
    g = load(path)
    o = oracle_mod.bilinear_align(g["txt"], g["vis"], g["tmask"], g["vmask"], np.float32, float(g["neg_inf"]),
                                  full=True, maxV=True, maxQ=True, diag=g["txt"].shape[0] == g["vis"].shape[0])
    ref = g["attmap"]
    masked = ref <= -1e19
    assert np.array_equal(o["full"][masked], ref[masked])
    assert np.abs(o["full"][~masked] - ref[~masked]).max() <= 2e-5
    assert np.array_equal(o["maxV"], o["full"].max(-1)) and np.array_equal(o["maxQ"], o["full"].max(-2))
    if o["diag"] is not None:
        idx = np.arange(ref.shape[0])
        assert np.array_equal(o["diag"], o["full"][idx, idx])


@pytest.mark.parametrize("path", golden_files("attnfuse_"), ids=golden_ids("attnfuse_"))
def test_oracle_attnfuse_matches_reference(oracle_mod, path):
    g = load(path)
    att, out = oracle_mod.attn_fuse(g["vis"], g["txt"], g["vis_mid"], g["enc_x"], g["ln_weight"], g["ln_bias"],
                                    float(g["ln_eps"]), np.float64)
    assert np.abs(att - g["attmap"]).max() <= 5e-6 and np.abs(out - g["out"]).max() <= 2e-5


@pytest.mark.parametrize("path", golden_files("attnfuse_"), ids=golden_ids("attnfuse_"))
def test_oracle_attnfuse_backward_matches_reference(oracle_mod, path):
    """Hand-derived adjoint vs torch autograd through the reference's own ops (joint.py:670-674)."""
    g = load(path)
    got = oracle_mod.attn_fuse_backward(g["vis"], g["txt"], g["vis_mid"], g["enc_x"], g["ln_weight"], g["dout"],
                                        float(g["ln_eps"]), np.float64)
    for name, arr in zip(("g_vis", "g_txt", "g_vis_mid", "g_enc_x", "g_ln_weight", "g_ln_bias"), got):
        ref = g[name]
        assert arr.shape == ref.shape, name
        assert np.abs(arr - ref).max() <= 2e-5 * max(1.0, np.abs(ref).max()), name   # the reference ran in fp32
    assert not got[1][:, 0].any()   # the root slot never enters the scores


@pytest.mark.parametrize("path", golden_files("rules_"), ids=golden_ids("rules_"))
def test_oracle_rules_matches_reference(oracle_mod, path):
    g = load(path)
    hm = g["head_mask"] if g["head_mask"].any() else None
    lz, gr, gd, groot = oracle_mod.dmv1o_rules(g["attach_rule"], g["dec"], g["root_rule"], g["token"], g["lengths"], hm,
                                               "log", np.float64)
    got_root = groot if int(g["root_per_sentence"]) else groot.sum(0, keepdims=True)
    assert np.abs(lz - g["logZ64"]).max() <= 1e-11 and np.abs(gr - g["grad_rule64"]).max() <= 1e-11
    assert np.abs(gd - g["grad_dec64"]).max() <= 1e-11 and np.abs(got_root - g["grad_root64"]).max() <= 1e-11


@pytest.mark.parametrize("path", golden_files("ground_"), ids=golden_ids("ground_"))
def test_oracle_grounding_loss_matches_reference(oracle_mod, path):
    """gather_logit_simple -> loss_grounding_factor_ce run from the reference's own methods (joint.py:406-419, 439-491)."""
    g = load(path)
    Q = g["txt"].shape[1]
    pen = seg = None
    if bool(g["use_pos_prior"]):
        pos_for = dict(obj=g["pos_for_obj"], rel=g["pos_for_rel"], attr=g["pos_for_attr"])
        pen, seg = oracle_mod.grounding_prior(g["tag"], g["factor_names"], g["vis_split"], pos_for, Q)
    o = oracle_mod.grounding_loss(g["txt"], g["vis"], g["tmask"], g["vmask"], g["marginal"], int(g["num_token"]),
                                  float(g["vis2txt_weight"]), pen, seg, float(g["neg_inf"]), np.float64)
    att = g["attmap_prior"]                       # the reference's attmap after the in-place prior
    assert np.allclose(o["maxV"], att.max(3), rtol=1e-5, atol=1e-5) and np.allclose(o["maxQ"], att.max(2), rtol=1e-5, atol=1e-5)
    assert abs(o["txt2vis"] - float(g["txt2vis_raw"])) <= 2e-5 * abs(float(g["txt2vis_raw"]))
    assert abs(o["vis2txt"] - float(g["vis2txt_raw"])) <= 2e-5 * abs(float(g["vis2txt_raw"]))
    want_total = float(g["total"])
    got_total = o["total"] if float(g["vis2txt_weight"]) > 0 else o["txt2vis"] / (o["txt2vis"] + 1e-6) * int(g["num_token"])
    assert abs(got_total - want_total) <= 1e-4 * abs(want_total)
    for name in ("g_txt", "g_vis"):
        assert np.abs(o[name] - g[name]).max() <= 2e-5 * max(1.0, np.abs(g[name]).max()), name


@pytest.mark.parametrize("path", golden_files("reduced_"), ids=golden_ids("reduced_"))
def test_oracle_gather_logit_reduced_matches_reference(oracle_mod, path):
    """gather_logit_reduced + the cross-entropy of loss_grounding_cap_img_ll, run from the reference's own methods
    (joint.py:421-432, 493-499), with autograd gradients."""
    g = load(path)
    o = oracle_mod.gather_logit_reduced(g["txt"], g["vis"], g["tmask"], g["vmask"], g["marginal"], g["g_logit"])
    assert np.allclose(o["logit"], g["logit"], rtol=2e-5, atol=2e-5)
    z = o["logit"] - o["logit"].max(1, keepdims=True)
    ce = -(z - np.log(np.exp(z).sum(1, keepdims=True)))[np.arange(len(z)), np.arange(len(z))].mean()
    assert abs(ce - float(g["loss"])) <= 1e-5 * max(1.0, abs(float(g["loss"])))
    for name in ("g_txt", "g_vis"):
        assert np.abs(o[name] - g[name]).max() <= 2e-5 * max(1.0, np.abs(g[name]).max()), name


@pytest.mark.parametrize("path", golden_files("gdecode_"), ids=golden_ids("gdecode_"))
def test_oracle_grounding_decode_matches_reference(oracle_mod, path):
    """gather_logit_simple -> decode_grounding_on_factor run from the reference's own methods (joint.py:406-419, 512-629)."""
    from conftest import gdecode_check_lists
    g = load(path)
    al = oracle_mod.bilinear_align(g["txt"], g["vis"], g["tmask"], g["vmask"], np.float32, -1e20, full=False, maxV=True, diag=True)
    assert np.allclose(al["diag"], g["diag_before"], rtol=1e-5, atol=1e-5) and np.allclose(al["maxV"], g["max_v"], rtol=1e-5, atol=1e-5)
    pos_for = dict(obj=g["pos_for_obj"], rel=g["pos_for_rel"], attr=g["pos_for_attr"])
    o = oracle_mod.grounding_decode(g["diag_before"], g["max_v"], g["tag"], g["factor_names"], g["vis_split"], pos_for,
                                    bool(g["use_pos_prior"]), bool(g["use_heuristic"]))
    assert np.array_equal(o["logit"], g["diag_after"])                                   # same fp32 edits, bit for bit
    assert np.array_equal(np.take_along_axis(o["logit"], o["top5"], -1), g["top_vals"])
    box_index = g["vis_box_index"] if g["vis_box_index"].size else None
    factor, img = oracle_mod.grounding_decode_lists(o["top5"], o["factor2img"], g["tmask"], g["factor_names"], g["vis_split"],
                                                    box_index)
    gdecode_check_lists(factor, img, g, g["diag_after"])


@pytest.mark.parametrize("path", golden_files("arcenc_"), ids=golden_ids("arcenc_"))
def test_oracle_arc_encoder_matches_reference(oracle_mod, path):
    """joint.py:281-287 (einsum + matmul + bias) and its autograd gradients."""
    from conftest import arcenc_check_w1_grad, arcenc_w1
    g = load(path)
    w1 = arcenc_w1(g)
    arc = oracle_mod.arc_encoder(g["child"], g["parent"], w1, g["w2"], g["b"], np.float64)
    assert np.abs(arc - g["arc"]).max() <= 2e-5 * max(1.0, np.abs(g["arc"]).max())
    d_child, d_parent, d_w1, d_w2, d_b = oracle_mod.arc_encoder_backward(g["child"], g["parent"], w1, g["w2"], g["dout"], np.float64)
    for name, arr in (("g_child", d_child), ("g_parent", d_parent), ("g_w2", d_w2), ("g_b", d_b)):
        assert np.abs(arr - g[name]).max() <= 2e-5 * max(1.0, np.abs(g[name]).max()), name
    arcenc_check_w1_grad(d_w1, g, 2e-5)


@pytest.mark.parametrize("path", golden_files("boxrel_"), ids=golden_ids("boxrel_"))
def test_oracle_box_rel_matches_reference(oracle_mod, path):
    """VisBoxRelSimpleEncoder.forward (box_rel.py:29-52), the reference's own module run by make_golden.py, and its autograd."""
    g = load(path)
    rel, g_feat, g_w, g_b = oracle_mod.box_rel(g["feat"], g["weight"], g["bias"], float(g["slope"]), True, g["dout"])
    for got, want in ((rel, g["rel"]), (g_feat, g["g_feat"]), (g_w, g["g_weight"]), (g_b, g["g_bias"])):
        assert got.shape == want.shape
        assert np.abs(got - want).max() <= 2e-5 * max(1.0, np.abs(want).max())     # fixtures are fp32 runs of the reference


# ------------------------------------------------------------------------------------------------ lang_feat_max_tree
@pytest.mark.parametrize("path", golden_files("langfeat_"), ids=golden_ids("langfeat_"))
def test_lang_feat_oracle_vs_reference(oracle_mod, path):
    """oracle.lang_feat / lang_feat_marginal / mlp restate joint.py:246-288 + nn/common.py:47-51; the fixtures are the
    reference's own `lang_feat_max_tree` (with its MLP modules and its DMV1o) and torch autograd through it (fp32)."""
    from conftest import arcenc_w1, arcenc_check_w1_grad
    g = load(path)
    w1 = arcenc_w1(g)
    heads, lengths = g["predicted"], g["lengths"]
    # the DP half: marginals from the oracle's inside-outside on the same merged potentials
    _, _, gatt = oracle_mod.dmv1o(g["merged_dec"], g["merged_attach"], lengths, "log", np.float64)
    _, _, vatt = oracle_mod.dmv1o(g["merged_dec"], g["merged_attach"], lengths, "max", np.float64)
    pred = np.zeros_like(heads)
    for b, h, c in zip(*np.nonzero(vatt.sum(-1))):
        pred[b, c] = h
    assert (pred == heads).all()
    marg, mask = oracle_mod.lang_feat_marginal(gatt, heads, lengths, bool(g["add_marginal"]))
    assert (mask == g["txt_mask"]).all()
    assert np.abs(marg - g["txt_marginal"]).max() <= 2e-5
    txt, grads = oracle_mod.lang_feat(g["x"], lengths, heads, g["w_word"], g["b_word"], g["w_child"], g["b_child"], g["w_parent"],
                                      g["b_parent"], w1, g["w2"], g["b_arc"], float(g["slope"]), g["dout"])
    assert np.abs(txt - g["txt"]).max() <= 2e-5 * max(1.0, np.abs(g["txt"]).max())
    for k in ("x", "w_word", "b_word", "w_child", "b_child", "w_parent", "b_parent", "w2", "b_arc"):
        ref = g["g_" + k]
        assert np.abs(grads[k] - ref).max() <= 5e-5 * max(1.0, np.abs(ref).max()), k
    arcenc_check_w1_grad(grads["w1"], g, 5e-5)


# ------------------------------------------------------------------------------------------------ scorer -> merged potentials
@pytest.mark.parametrize("path", golden_files("scorer_"), ids=golden_ids("scorer_"))
def test_ndmv_potentials_oracle_vs_reference(oracle_mod, path):
    """oracle.ndmv_potentials restates ldndmv.py:184-209 + nn/dmv_spec.py:66-76; the fixtures are the reference's own
    `DiscriminativeNDMV._forward` (its scorer modules executed) and torch autograd through it (fp32)."""
    g = load(path)
    md, ma, grads = oracle_mod.ndmv_potentials(g["x1"], g["x2"], g["y1"], g["y2"], g["root_rule"], g["token"], g["head_mask"],
                                               float(g["mask_fill"]), g["g_merged_dec"], g["g_merged_attach"])
    big = np.abs(g["merged_attach"]) > 1e11                                 # -1e12 / -1e20 fills: exact
    assert (ma[big].astype(np.float32) == g["merged_attach"][big]).all()
    assert np.abs(ma[~big] - g["merged_attach"][~big]).max() <= 1e-5
    bigd = np.abs(g["merged_dec"]) > 1e11
    assert (md[bigd].astype(np.float32) == g["merged_dec"][bigd]).all() and np.abs(md[~bigd] - g["merged_dec"][~bigd]).max() <= 1e-5
    for k in ("x1", "x2", "y1", "y2", "root_rule"):
        ref = g["g_" + k]
        assert np.abs(grads[k] - ref).max() <= 2e-5 * max(1.0, np.abs(ref).max()), k
