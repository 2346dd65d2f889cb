"""CPU: the C-ABI shared library loads without a GPU, exports every symbol include/vlgae_amd.h declares,
and rejects bad arguments on the host side (no compute calls are made here)."""
import ctypes
import os
import re

import pytest

from conftest import ROOT


@pytest.fixture(scope="module")
def lib():
    from vlgae_amd.build import build_library
    build_library()            # hipcc cross-compiles for gfx950 without a GPU
    from vlgae_amd import _C
    return _C.lib()


def declared_symbols():
    text = open(os.path.join(ROOT, "include", "vlgae_amd.h")).read()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    return sorted(set(re.findall(r"\b(vlg_[a-z0-9_]+)\s*\(", text)))


def test_header_symbols_are_exported(lib):
    syms = declared_symbols()
    assert len(syms) >= 11, syms
    for s in syms:
        assert hasattr(lib, s), f"{s} declared in include/vlgae_amd.h but not exported by the library"


def test_binding_table_matches_header(lib):
    from vlgae_amd import _C
    assert sorted(_C.SIGNATURES) == declared_symbols()
    # argument counts agree with the prototypes
    text = re.sub(r"/\*.*?\*/", "", open(os.path.join(ROOT, "include", "vlgae_amd.h")).read(), flags=re.S)
    for name, (_, args) in _C.SIGNATURES.items():
        m = re.search(name + r"\s*\(([^)]*)\)", text)
        params = [p for p in m.group(1).split(",") if p.strip() and p.strip() != "void"]
        assert len(params) == len(args), (name, len(params), len(args))


def test_version_and_error_plumbing(lib):
    from vlgae_amd import _C
    assert lib.vlg_version() >= 100
    # shape / dtype / semiring / null-buffer validation happens before any HIP call
    rc = lib.vlg_dmv1o_inside(None, None, None, 4, 1, 0, 0, None, None, 0, None)
    assert rc == 0x1001 and b"N >= 2" in lib.vlg_last_error()
    rc = lib.vlg_dmv1o_inside(None, None, None, 4, 300, 0, 0, None, None, 0, None)
    assert rc == 0x1001 and b"255" in lib.vlg_last_error()
    assert lib.vlg_dmv1o_inside(None, None, None, 4, 8, 7, 0, None, None, 0, None) == 0x1002
    assert lib.vlg_dmv1o_inside(None, None, None, 4, 8, 0, 5, None, None, 0, None) == 0x1003
    assert lib.vlg_dmv1o_inside(None, None, None, 4, 8, 0, 0, None, None, 0, None) == 0x1003      # null buffers
    assert lib.vlg_deptree_inside_outside(None, None, 2, 8, 0, 0, None, None, None, None, 0, None) == 0x1003
    assert lib.vlg_bilinear_align(None, None, None, None, 2, 3, 4, 5, 8, 0, -1e20, None, None, None,
                                  ctypes.c_void_p(16), None) == 0x1001                               # diag needs A == B
    assert lib.vlg_dmv1o_merge(None, None, None, 2, 0, 0, 0.0, -1e12, None, None, None) == 0x1001
    # a head vector only exists for the Viterbi tree: heads with the Log semiring is an argument error, before any launch
    one = ctypes.c_void_p(16)
    assert lib.vlg_dmv1o_rules(one, one, one, 0, one, None, one, 2, 5, 7, 0, 0, -1e20, None, one, None, None, None, one,
                               None, 0, None) == 0x1003 and b"VLG_SEMIRING_MAX" in lib.vlg_last_error()
    assert lib.vlg_bilinear_align_backward(one, one, one, None, None, 2, 2, 5, 5, 48, 0, None, 0, one, one, None) == 0x1001   # d not in {32,64,128}
    assert lib.vlg_bilinear_align_backward_workspace(4, 4, 82, 36, 128, 1) == (128 * (4 * 36 + 96) + 4 * 128 * 96) * 2 and lib.vlg_bilinear_align_backward_workspace(4, 4, 82, 36, 64, 0) == 0 and lib.vlg_bilinear_align_backward_workspace(4, 4, 82, 36, 128, 0) == 2 * (4 * 128 * 64 + 4 * 128 * 96) * 2
    assert lib.vlg_bilinear_align_backward(one, one, one, None, None, 4, 4, 82, 36, 128, 1, None, 0, one, one, None) == 0x1004   # fast path without its scratch
    assert lib.vlg_dmv1o_count_sum(None, None, 4, 8, None, None) == 0x1003
    # round 4: the language-side kernels take the activations' storage type and the SharedDropout masks
    assert lib.vlg_version() >= 130
    assert lib.vlg_langfeat_root_cat(one, one, 2, 5, 64, 0, one, 7, None) == 0x1002                               # out_dtype
    assert lib.vlg_langfeat_split(one, one, one, 8, 2, 6, 32, 0, 0.01, one, one, one, None, None) == 0x1001       # ld_drop < 3 d
    assert lib.vlg_langfeat_split(one, one, None, 0, 2, 6, 32, 5, 0.01, one, one, one, None, None) == 0x1002      # act_dtype
    assert lib.vlg_langfeat_rowscale(one, 36, one, 2, 6, 36, 36, 0, one, 36, 36, None) == 0x1001                  # d % 8
    assert lib.vlg_langfeat_rowscale(one, 32, one, 2, 6, 32, 32, 0, one, 32, 24, None) == 0x1001                  # width < d
    assert lib.vlg_langfeat_rowscale(None, 32, one, 2, 6, 32, 32, 0, one, 32, 32, None) == 0x1003                 # null input
    assert lib.vlg_langfeat_arc_out(one, None, 2, 6, 32, 9, one, None) == 0x1002
    assert lib.vlg_linear_wgrad_workspace(4096, 24, 72) > 0 and lib.vlg_linear_wgrad_workspace(4096, 20, 72) == 0     # multiples of 8
    with pytest.raises(RuntimeError, match="N >= 2"):
        _C.check(lib.vlg_dmv1o_inside(None, None, None, 4, 1, 0, 0, None, None, 0, None), "dmv1o_inside")
    # empty batches are a no-op success
    assert lib.vlg_dmv1o_inside(None, None, None, 0, 8, 0, 0, None, None, 0, None) == 0
    assert lib.vlg_deptree_inside(None, None, 0, 8, 0, 0, None, None, 0, None) == 0


def test_workspace_sizing(lib):
    from vlgae_amd import _C
    # charts fit the 160 KiB LDS -> no workspace; long sentences spill ("hard parts")
    assert lib.vlg_workspace_bytes(_C.OP_DMV1O_INSIDE_OUTSIDE, 256, 41, 0) == 0
    assert lib.vlg_workspace_bytes(_C.OP_DMV1O_INSIDE, 256, 81, 0) == 0
    w81 = lib.vlg_workspace_bytes(_C.OP_DMV1O_INSIDE_OUTSIDE, 256, 81, 0)
    assert w81 > 0 and w81 % 256 == 0
    assert lib.vlg_workspace_bytes(_C.OP_DMV1O_INSIDE_OUTSIDE, 512, 81, 0) == 2 * w81
    # the Max semiring's outside pass is the back-pointer walk (round 4): its lean layout spills only gI at N = 81 -- the query must size
    # THAT layout, not the unreachable one-hot replay
    wmax = lib.vlg_workspace_bytes(_C.OP_DMV1O_INSIDE_OUTSIDE, 256, 81, 1)
    assert 0 < wmax < w81 and lib.vlg_workspace_bytes(_C.OP_DMV1O_INSIDE_OUTSIDE, 256, 41, 1) == 0
    assert lib.vlg_workspace_bytes(_C.OP_DEPTREE_INSIDE_OUTSIDE, 256, 81, 0) == 0
    assert lib.vlg_workspace_bytes(_C.OP_DEPTREE_INSIDE_OUTSIDE, 4, 200, 0) > 0
    assert lib.vlg_workspace_bytes(99, 4, 41, 0) == 0 and lib.vlg_workspace_bytes(0, 0, 41, 0) == 0


def test_attn_and_grounding_entry_points_validate_on_the_host(lib):
    """Same contract for the later entry points: shape / dtype / null / workspace errors before any HIP call."""
    P = ctypes.c_void_p
    one = P(16)   # any non-null pointer: validation must fail before it is dereferenced
    # attention-fuse adjoint: d, h multiples of 16 and <= 256
    bw = lib.vlg_attn_fuse_backward
    tail = (None, one, 1 << 30, one, one, one, one, one, one, None)
    assert bw(one, one, one, one, one, one, 5 * 64, 64, 2, 5, 7, 24, 64, 0, 1e-5, 0, 0, *tail) == 0x1001
    assert b"multiples of 16" in lib.vlg_last_error()
    assert bw(one, one, one, one, one, one, 5 * 64, 64, 2, 5, 7, 32, 512, 0, 1e-5, 0, 0, *tail) == 0x1001
    assert bw(one, one, one, one, one, one, 5 * 64, 64, 2, 5, 7, 32, 64, 9, 1e-5, 0, 0, *tail) == 0x1002
    assert bw(one, one, one, one, one, one, 5 * 64, 64, 2, 5, 7, 32, 64, 0, 1e-5, 0, 1, *tail) == 0x1002      # bf16 gradients of fp32 inputs
    assert b"grad_dtype" in lib.vlg_last_error()
    assert bw(None, one, one, one, one, one, 5 * 64, 64, 2, 5, 7, 32, 64, 0, 1e-5, 0, 0, *tail) == 0x1003
    assert bw(one, one, one, one, one, one, 64, 2, 2, 5, 7, 32, 64, 0, 1e-5, 0, 0, *tail) == 0x1001   # dout stride
    wsq = lib.vlg_attn_fuse_backward_workspace
    need = wsq(2, 5, 7, 32, 64, 0, 0)
    assert need > 0 and wsq(4, 5, 7, 32, 64, 0, 0) > need and wsq(0, 5, 7, 32, 64, 0, 0) == 0
    assert wsq(2, 5, 7, 32, 64, 1, 0) > need                        # bf16 gradients: the fp32 dy rows live in the scratch
    assert wsq(2, 5, 300, 32, 64, 0, 0) > wsq(2, 5, 300, 32, 64, 0, 300)   # above 256 keys: chunk records of the key-split form
    assert bw(one, one, one, one, one, one, 5 * 64, 64, 2, 5, 7, 32, 64, 0, 1e-5, 0, 0, None, one, need - 1, one, one, one, one, one, one, None) == 0x1004
    assert b"workspace" in lib.vlg_last_error()
    # grounding loss: d in {32, 64, 128}, 16-bit positions, prior table needs its segment map
    gl = lib.vlg_grounding_loss
    need = lib.vlg_grounding_loss_workspace(3, 14, 9)
    assert need > 0 and need % 256 == 0 and lib.vlg_grounding_loss_workspace(6, 14, 9) > 2 * need
    args = lambda **kw: [kw.get("txt", one), one, None, None, one, kw.get("pen", None), kw.get("seg", None), kw.get("n_seg", 0), 3,
                         kw.get("Q", 14), 9, kw.get("d", 32), kw.get("dt", 0), -1e20, 30.0, 1.0, one, kw.get("ws", need), one, None,
                         None, None]
    assert gl(*args(d=48)) == 0x1001 and b"d=48" in lib.vlg_last_error()
    assert gl(*args(d=512)) == 0x1001
    assert gl(*args(Q=70000)) == 0x1001
    assert gl(*args(dt=3)) == 0x1002
    assert gl(*args(txt=None)) == 0x1003
    assert gl(*args(pen=one)) == 0x1003 and b"segments" in lib.vlg_last_error()
    assert gl(*args(ws=need - 1)) == 0x1004
    # attention-fuse forward
    assert lib.vlg_attn_fuse(one, one, one, one, one, one, 2, 0, 7, 32, 64, 0, 1e-5, 0, None, 0, None, None, one, None) == 0x1001
    assert lib.vlg_attn_fuse(one, one, one, one, one, one, 0, 5, 7, 32, 64, 0, 1e-5, 0, None, 0, None, None, one, None) == 0      # empty batch
    # key-split form: workspace only above 256 keys (or when a chunk size is forced), and it is checked before any launch
    fq = lib.vlg_attn_fuse_workspace
    assert fq(2, 5, 7, 64, 0) == 0 and fq(2, 5, 256, 64, 0) == 0 and fq(2, 5, 257, 64, 0) > 0 and fq(2, 5, 130, 64, 64) > 0
    rec = 4 * (16 * 256 + 32)                                             # one record: [16 channel tiles][64 lanes][4] + max[16] + sum[16] floats
    chunks = lambda B, L: fq(B, L, 1369, 256, 0) // rec // (B * ((L + 15) // 16)) - 1   # (+ one merged record per (sentence, word tile))
    assert fq(64, 40, 1369, 256, 0) % (64 * 3 * rec) == 0 and 4 <= chunks(64, 40) <= 11   # B = 64: a few 64-key steps per chunk
    assert chunks(256, 40) < chunks(64, 40)                                            # a big batch needs few chunks per sentence (or none: -1)
    assert chunks(2, 5) == 22                                                          # a small one gets 64-key chunks
    assert fq(2, 5, 1369, 256, 128) == 2 * 1 * (11 + 1) * rec                          # forced: 128 keys per chunk
    assert lib.vlg_attn_fuse(one, one, one, one, one, one, 2, 5, 300, 32, 64, 0, 1e-5, 0, one, 16, None, None, one, None) == 0x1004
    assert lib.vlg_attn_fuse_saved_bytes(2, 5, 300, 64, 0) == 2 * 1 * 4 * (4 * 256 + 32) and lib.vlg_attn_fuse_saved_bytes(2, 5, 36, 64, 0) == 0


def test_align_reduced_validates_on_the_host(lib):
    P = ctypes.c_void_p
    one = P(16)
    fw, bw = lib.vlg_align_reduced, lib.vlg_align_reduced_backward
    need = lib.vlg_align_reduced_workspace(3, 14)
    assert need > 0 and need % 256 == 0 and lib.vlg_align_reduced_workspace(0, 14) == 0
    assert fw(one, one, None, None, one, 3, 14, 9, 48, 0, -1e20, one, need, one, None) == 0x1001 and b"d=48" in lib.vlg_last_error()
    assert fw(one, one, None, None, one, 3, 14, 9, 32, 5, -1e20, one, need, one, None) == 0x1002
    assert fw(None, one, None, None, one, 3, 14, 9, 32, 0, -1e20, one, need, one, None) == 0x1003
    assert fw(one, one, None, None, one, 3, 14, 9, 32, 0, -1e20, one, need - 1, one, None) == 0x1004
    assert fw(one, one, None, None, one, 3, 14, 9, 32, 0, -1e20, one, need, None, None) == 0x1003
    assert bw(one, one, None, None, one, one, 3, 14, 9, 32, 0, one, need, None, None, None) == 0x1003
    assert bw(one, one, None, None, one, one, 3, 70000, 9, 32, 0, one, need, one, one, None) == 0x1001


def test_grounding_decode_validates_on_the_host(lib):
    P = ctypes.c_void_p
    one = P(16)
    dec = lib.vlg_grounding_decode
    ok = lambda **kw: [kw.get("logit", one), kw.get("pen", None), kw.get("seg", None), kw.get("n_seg", 0), kw.get("B", 2), 14,
                       kw.get("V", 40), kw.get("heur", 1), kw.get("n_box", 5), kw.get("rel", 5), kw.get("attr", 30), 7,
                       kw.get("maxV", None), 2, kw.get("f2i", None), kw.get("top", one), None, 0, None]
    assert lib.vlg_grounding_decode_workspace(4, 36) == 512 and lib.vlg_grounding_decode_workspace(0, 36) == 0
    assert dec(*ok(B=0)) == 0                                           # empty batch: nothing to do
    assert dec(*ok(V=0)) == 0x1001
    assert dec(*ok(logit=None)) == 0x1003 and dec(*ok(top=None)) == 0x1003
    assert dec(*ok(pen=one)) == 0x1003 and b"seg_of_v" in lib.vlg_last_error()
    assert dec(*ok(maxV=one)) == 0x1003 and b"factor2img" in lib.vlg_last_error()
    assert dec(*ok(n_box=0)) == 0x1001
    assert dec(*ok(rel=20)) == 0x1001 and b"relation block" in lib.vlg_last_error()      # 20 + 5^2 > 40
    assert dec(*ok(attr=36)) == 0x1001 and b"attribute block" in lib.vlg_last_error()    # 36 + 5 > 40


def test_trilinear_entry_points_validate_on_the_host(lib):
    P = ctypes.c_void_p
    one = P(16)
    tri, bwd = lib.vlg_trilinear, lib.vlg_trilinear_backward
    assert tri(one, one, one, 10, 32, 24, 32, 0, one, None) == 0x1001 and b"multiple of 16" in lib.vlg_last_error()
    assert tri(one, one, one, 10, 32, 256, 32, 0, one, None) == 0x1001
    assert tri(one, one, one, 10, 32, 32, 48, 0, one, None) == 0x1001 and b"contracted dimension 48" in lib.vlg_last_error()
    assert tri(one, one, one, 10, 32, 32, 32, 2, one, None) == 0x1002
    assert tri(None, one, one, 10, 32, 32, 32, 0, one, None) == 0x1003
    assert tri(None, None, None, 0, 32, 32, 32, 0, None, None) == 0                  # no rows: nothing to do
    need = lib.vlg_trilinear_backward_workspace(100, 32, 64, 32, 1)
    assert need > 0 and need % 256 == 0 and lib.vlg_trilinear_backward_workspace(100, 32, 64, 32, 0) > need   # fp32 copies are larger
    assert bwd(one, one, one, one, 100, 32, 64, 32, 1, one, need - 1, one, one, one, None) == 0x1004
    assert bwd(one, one, one, one, 100, 32, 48, 32, 1, one, need, one, one, one, None) == 0x1001
    assert bwd(one, one, one, None, 100, 32, 64, 32, 1, one, need, one, one, one, None) == 0x1003


def test_round4_entry_points_validate_on_the_host(lib):
    """vlg_ff_* (element-wise passes of the parser's feed-forwards), vlg_dmv1o_marginals_viterbi and the strided score construction:
    shape / dtype / alignment / null errors before any HIP call."""
    P = ctypes.c_void_p
    one, odd = P(64), P(68)
    f = ctypes.c_float(0.01)
    assert lib.vlg_ff_act(one, None, None, f, None, 0, 0.0, one, 10, 1, 12, 0, 1, f, None) == 0x1001                 # H not a multiple of 8
    assert b"multiple of 8" in lib.vlg_last_error()
    assert lib.vlg_ff_act(one, None, None, f, None, 0, 0.0, one, 10, 2, 64, 1, 1, f, None) == 0x1001                 # the permutation is of J = 4
    assert lib.vlg_ff_act(one, None, None, f, None, 0, 0.0, one, 10, 4, 64, 1, 1, f, None) == 0x1003                 # ... and not in place
    assert lib.vlg_ff_act(one, None, None, f, None, 0, 0.0, one, 10, 1, 64, 0, 5, f, None) == 0x1002
    assert lib.vlg_ff_act(odd, None, None, f, None, 0, 0.0, one, 10, 1, 64, 0, 1, f, None) == 0x1003 and b"aligned" in lib.vlg_last_error()
    assert lib.vlg_ff_act(one, None, None, f, None, 0, 0.0, one, 0, 1, 64, 0, 1, f, None) == 0                       # nothing to do
    assert lib.vlg_ff_act_backward(one, one, None, f, None, 0, 0.0, one, None, 10, 4, 64, 1, 0, 1, f, None) == 0x1003
    assert lib.vlg_ff_act_backward(one, None, None, f, None, 0, 0.0, one, None, 10, 1, 64, 0, 0, 1, f, None) == 0x1003
    assert lib.vlg_ff_mlp_act(one, None, None, None, 2, 5, 3, 64, 1, f, None) == 0x1003             # parent rows need the context term
    assert lib.vlg_ff_mlp_act(one, one, None, None, 2, 0, 3, 64, 1, f, None) == 0x1001
    assert lib.vlg_ff_mlp_act_backward(None, None, one, None, None, one, 2, 5, 3, 64, 1, f, None) == 0x1003
    assert lib.vlg_dmv1o_marginals_viterbi_supported(41) == 1 and lib.vlg_dmv1o_marginals_viterbi_supported(81) == 0
    assert lib.vlg_dmv1o_marginals_viterbi_supported(1) == 0 and lib.vlg_dmv1o_marginals_viterbi_supported(300) == 0
    mv = lib.vlg_dmv1o_marginals_viterbi
    assert mv(one, one, one, 4, 81, 0, one, None, one, one, None, None, one, None) == 0x1001 and b"two streams" in lib.vlg_last_error()
    assert mv(one, one, one, 4, 41, 7, one, None, one, one, None, None, one, None) == 0x1002
    assert mv(one, one, one, 4, 41, 0, one, None, None, one, None, None, one, None) == 0x1003        # the marginals are not optional
    assert mv(one, one, one, 4, 41, 0, one, None, one, one, one, None, one, None) == 0x1003         # tree_dec needs tree_attach
    assert mv(one, one, one, 0, 41, 0, one, None, one, one, None, None, one, None) == 0
    # score construction: row strides below the rank are refused
    sc = lib.vlg_ndmv_potentials
    assert sc(one, 8, one, 16, one, 16, one, 16, one, one, None, 2, 5, 7, 16, 0, ctypes.c_float(-1e20), 0, one, one, None) == 0x1001
    assert b"row strides" in lib.vlg_last_error()
    assert lib.vlg_linear_wgrad(one, 64, one, 64, 4096, 64, 64, 1, one, 1 << 30, 9, one, 64, one, None, None) == 0x1002
    sg = lambda **kw: lib.vlg_small_gemm(kw.get("a", one), 0, 16, 1, one, 0, 8, 1, kw.get("c", one), 0, kw.get("ldc", 8), None, 0, kw.get("u", None), 0, None, 0,
                                         kw.get("batch", 1), kw.get("M", 4), 8, 16, ctypes.c_float(1.0), 0, kw.get("dt", 1), 1, None)
    assert sg(M=0) == 0x1001 and sg(ldc=4) == 0x1001 and sg(dt=5) == 0x1002 and sg(a=None) == 0x1003 and sg(batch=0) == 0
    assert sg(u=one) == 0x1003 and b"rank-one" in lib.vlg_last_error()
    assert lib.vlg_langfeat_root_cat_backward(one, one, 2, 5, 64, 0, one, 9, None) == 0x1002
