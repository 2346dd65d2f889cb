"""CPU: host-side mirror of the reference's interface -- names, constants, shapes, error behaviour --
and the rule that the product has no CPU / oracle fallback."""
import os
import re

import pytest
import torch

from conftest import ROOT


def test_api_surface_matches_reference_names():
    import vlgae_amd.torch_struct as ts
    # src/model/torch_struct/dmv.py:7-15
    assert (ts.NOCHILD, ts.HASCHILD, ts.LEFT, ts.RIGHT, ts.GO, ts.STOP) == (1, 0, 0, 1, 0, 1)
    assert (ts.DIR_NUM, ts.VAL_NUM, ts.DEC_NUM) == (2, 2, 2)
    assert ts.NEGINF == -1e12 and ts.LogSemiring.zero == -1e12 and ts.MaxSemiring.zero == -1e12
    for cls in (ts.DMV1o, ts.DependencyCRF):
        assert issubclass(cls, ts.StructDistribution)
        for prop in ("partition", "max", "argmax", "marginals", "mode"):
            assert hasattr(cls, prop)
    assert callable(ts.DMV1o.merge)
    from vlgae_amd import align
    import inspect
    assert list(inspect.signature(align.gather_logit_simple).parameters) == ["self", "inputs", "vis", "txt", "vp"]


def test_no_cpu_fallback():
    """CPU tensors must raise, not silently compute somewhere else."""
    import vlgae_amd.torch_struct as ts
    from vlgae_amd import align
    dec, attach = torch.zeros(2, 5, 2, 2, 2), torch.zeros(2, 5, 5, 2)
    lengths = torch.tensor([4, 3])
    with pytest.raises(RuntimeError, match="no CPU path"):
        ts.DMV1o([dec, attach], lengths).partition
    with pytest.raises(RuntimeError, match="no CPU path"):
        ts.DMV1o([dec, attach], lengths).argmax
    with pytest.raises(RuntimeError, match="no CPU path"):
        ts.DependencyCRF(torch.zeros(2, 5, 5), lengths).marginals
    with pytest.raises(RuntimeError, match="no CPU path"):
        ts.DMV1o.merge(torch.zeros(2, 4, 2, 2, 2), torch.zeros(2, 4, 4, 2), torch.zeros(2, 4))
    with pytest.raises(RuntimeError, match="no CPU path"):
        align.bilinear_align(torch.zeros(2, 3, 8), torch.zeros(2, 4, 8))
    with pytest.raises(RuntimeError, match="no CPU path"):
        align.attention_fuse(torch.zeros(2, 4, 8), torch.zeros(2, 4, 8), torch.zeros(2, 4, 6), torch.zeros(2, 3, 6),
                             torch.ones(6), torch.zeros(6))


def test_product_never_imports_the_oracle():
    pat = re.compile(r"^\s*(from|import)\s+(oracle|tests)\b|cpu_oracle|libvlg_oracle|libvlg_emu", re.M)
    for dirpath, _, files in os.walk(os.path.join(ROOT, "vlgae_amd")):
        for f in files:
            if f.endswith((".py", ".hip", ".cpp", ".h")):
                text = open(os.path.join(dirpath, f)).read()
                assert not pat.search(text), f"{f} references test infrastructure"


def test_missing_library_fails_loudly(monkeypatch, tmp_path):
    from vlgae_amd import _C
    monkeypatch.setattr(_C, "_lib", None)
    monkeypatch.setattr(_C, "LIB_PATH", str(tmp_path / "nope.so"))
    with pytest.raises(RuntimeError, match="no CPU fallback"):
        _C.lib()


def test_out_of_scope_api_raises():
    import vlgae_amd.torch_struct as ts
    d = ts.DependencyCRF(torch.zeros(1, 4, 4))
    for call in (lambda: d.entropy, lambda: d.kmax(2), lambda: d.topk(2), lambda: d.sample((1,)), lambda: d.count,
                 lambda: d.kl(d), lambda: d.cross_entropy(d), lambda: d.risk(None), lambda: d.gumbel_crf(),
                 lambda: d.enumerate_support()):
        with pytest.raises(NotImplementedError):
            call()
    with pytest.raises(AssertionError):
        ts.DependencyCRF(torch.zeros(1, 4, 4), multiroot=True)       # deptree.py:26-27


def test_distribution_metadata():
    import vlgae_amd.torch_struct as ts
    dec, attach = torch.zeros(3, 6, 2, 2, 2), torch.zeros(3, 6, 6, 2)
    d = ts.DMV1o([dec, attach], torch.tensor([5, 4, 3]))
    assert d.batch_shape == torch.Size([3]) and d.event_shape == torch.Size([6, 2, 2, 2])   # distributions.py:45-53,248-251
    assert d.log_potentials[1] is attach
    c = ts.DependencyCRF(torch.zeros(3, 6, 6))
    assert c.batch_shape == torch.Size([3]) and c.event_shape == torch.Size([6, 6]) and c.lengths is None


def test_shard_bounds():
    from vlgae_amd.dist import shard_bounds
    for n in (0, 1, 7, 256, 2048, 2049):
        for world in (1, 2, 3, 8):
            spans = [shard_bounds(n, r, world) for r in range(world)]
            assert spans[0][0] == 0 and spans[-1][1] == n
            assert all(spans[i][1] == spans[i + 1][0] for i in range(world - 1))
            sizes = [e - s for s, e in spans]
            assert max(sizes) - min(sizes) <= 1


def test_bench_train_shapes_match_the_step_and_parser_ff_names():
    """The dry-run stand-in of the sharded training step (vlgae_amd.bench.sharded_step._param_shapes) must list exactly the trainable
    leaves of vlgae_amd.train_step.build with their shapes, and vlgae_amd.parser_ff.param_names must cover every "ff." parameter --
    otherwise the gloo plumbing tests would exercise a different flat-gradient layout than the GPU run."""
    import torch
    from vlgae_amd import parser_ff, train_step
    from vlgae_amd.bench import sharded_step as bench_train
    f = bench_train.FF_SHAPES
    P = train_step.init_feed_forward(torch.Generator().manual_seed(0), torch.device("cpu"), torch.float32, f["E"], f["h"], f["Et"], f["T"],
                                     f["H"], f["nb"], f["r"])
    shapes, groups = bench_train._param_shapes()
    ff_keys = {k for k in shapes if k.startswith("ff.") or k in ("token_emb", "root_emb", "dec_emb")}
    assert ff_keys == set(P), ff_keys ^ set(P)
    assert all(tuple(P[k].shape) == tuple(shapes[k]) for k in P)
    assert set(parser_ff.param_names(f["nb"])) == {k for k in P if k.startswith("ff.")}
    assert set(parser_ff.param_names(0)) == {k for k in train_step.init_feed_forward(torch.Generator().manual_seed(0), torch.device("cpu"),
                                                                                    torch.float32, 8, 8, 8, 3, 8, 0, 4) if k.startswith("ff.")}
    assert sorted(k for g in groups for k in g) == sorted(shapes)          # the readiness groups partition the trainable leaves
    assert groups[0] == ["w1", "w2", "b"] and "w_vis" in groups[2] and groups[3] == ["w_text", "w_venc", "b_venc"]
    # the two encoders in front of the step: MLPEncoder.linear [h, E] without bias, the visual encoder's stacked [F h, 2 n] / [F h]
    assert shapes["w_text"] == (f["h"], f["E"]) and shapes["w_venc"] == (f["h"], 2 * bench_train.N_VIS) and shapes["b_venc"] == (f["h"],)
    # every float of the shipped model's path is in the all-reduced buffer: 6.48 M parameters = 25.9 MB (SURVEY.md section 8e says ~7 M)
    n_model = sum(int(torch.Size(s).numel()) for s in bench_train._param_shapes(n_enc=3)[0].values())
    assert 6.3e6 < n_model < 7.5e6, n_model


def test_parser_ff_refuses_unsupported_widths():
    """n_mid != hidden_size or unequal scorer ranks are valid reference configurations that the fused pass does not cover --
    it must refuse them by shape (no GPU needed: the check runs before any kernel)."""
    import pytest
    import torch
    from vlgae_amd import parser_ff, train_step
    P = train_step.init_feed_forward(torch.Generator().manual_seed(0), torch.device("cpu"), torch.float32, 16, 8, 8, 3, 8, 0, 4)
    P = {k: v.detach() for k, v in P.items()}
    parser_ff._validate_shapes(P, 0, 8, 24)
    bad = dict(P)
    bad["ff.mid_ff.linear1.weight"] = torch.zeros(12, 8)                     # n_mid = 12 != hidden_size
    with pytest.raises(ValueError, match="n_mid == hidden_size"):
        parser_ff._validate_shapes(bad, 0, 8, 24)
    bad = dict(P)
    bad["ff.dec_scorer.project1.weight"] = torch.zeros(6, 8)                 # dec_rank != attach_rank
    with pytest.raises(ValueError, match="one rank shared"):
        parser_ff._validate_shapes(bad, 0, 8, 24)


def test_drop_in_import_selects_backward_on_the_calling_thread():
    """Importing vlgae_amd.torch_struct (what the reference's import alias does) keeps autograd's backward on the calling thread unless
    VLGAE_AMD_AUTOGRAD_THREAD=engine (checked in child interpreters: the setting is process-global)."""
    import subprocess
    import sys
    from conftest import ROOT
    code = "import torch, vlgae_amd.torch_struct; print(torch.autograd.is_multithreading_enabled())"
    env = dict(os.environ, PYTHONPATH=ROOT)
    env.pop("VLGAE_AMD_AUTOGRAD_THREAD", None)
    assert subprocess.run([sys.executable, "-c", code], env=env, stdout=subprocess.PIPE, text=True, check=True).stdout.strip() == "False"
    env["VLGAE_AMD_AUTOGRAD_THREAD"] = "engine"
    assert subprocess.run([sys.executable, "-c", code], env=env, stdout=subprocess.PIPE, text=True, check=True).stdout.strip() == "True"


def test_factor_layout_and_mask_match_the_reference_made_fixtures():
    """`encoders.factor_layout` / `factor_mask` (the layout and mask `vis_feat_unprune` builds, joint.py:140-171) against the three trainstep
    fixtures, whose `vis_mask`, `vis_split` and `factor_names` were produced by the reference's own method: object | relation (strict upper
    triangle of the box-mask outer product) | attribute | image rows, in that order.  Host logic only: runs without a GPU."""
    import glob
    import os
    import numpy as np
    import torch
    from vlgae_amd import encoders
    files = sorted(glob.glob(os.path.join(os.path.dirname(__file__), "golden", "trainstep_*.npz")))
    assert len(files) == 3
    seen = set()
    for f in files:
        g = np.load(f, allow_pickle=True)
        names = [str(n) for n in g["factor_names"]]
        add = dict(add_rel="rel" in names, add_attr="attr" in names, add_image="img" in names)
        box_mask = torch.from_numpy(g["box_mask"]).bool()
        off, V, split, fn = encoders.factor_layout(box_mask.shape[1], **add)
        assert fn == names and split == [int(x) for x in g["vis_split"]] and V == g["vis_mask"].shape[1] == sum(split)
        assert off["box"] == 0 and all((off[k] >= 0) == v for k, v in (("rel", add["add_rel"]), ("attr", add["add_attr"]), ("img", add["add_image"])))
        assert np.array_equal(encoders.factor_mask(box_mask, **add).numpy(), g["vis_mask"].astype(bool))
        seen.add(tuple(names))
    assert len(seen) == 3   # obj | obj rel attr | obj rel attr img
