"""Data feed: vlgae_amd.feed -> C ABI (csrc/vlg_feed.cpp), host code.

Parity: the golden fixtures are outputs of the reference's own ConstantTokenNumSampler and _COCODetFeatLazyLoader
(tests/golden/make_golden.py:feed_cases); batches and tensors must be identical.  Random cases go against the oracle's
dense restatement (oracle/cpu_oracle.py:feed_kmeans / feed_batches).
"""
import ctypes
import os

import numpy as np
import pytest
import torch

from conftest import golden_files, golden_ids, load


def _uncsr(off, items):
    return [items[off[j]:off[j + 1]].tolist() for j in range(len(off) - 1)]


def _sampler(g, **kw):
    from vlgae_amd.feed import ConstantTokenNumSampler
    torch.manual_seed(int(g["torch_seed"]))
    return ConstantTokenNumSampler(g["seq_len"].tolist(), int(g["max_token"]), int(g["max_sentence"]), int(g["num_bucket"]),
                                   int(g["single_sent_threshold"]), bool(g["sort_in_batch"]), bool(g["shuffle"]),
                                   bool(g["force_same_len"]), **kw)


@pytest.mark.parametrize("path", golden_files("feed_sampler_"), ids=golden_ids("feed_sampler_"))
def test_sampler_matches_reference(path):
    g = load(path)
    sm = _sampler(g)
    assert np.array_equal(np.asarray(sm.sizes, np.float64), g["sizes"])              # centroids: bit-equal float32 values
    assert sm.buckets == _uncsr(g["bucket_offsets"], g["bucket_items"])
    assert sm.chunks == g["chunks"].tolist()
    for e in range(3):
        assert list(sm) == _uncsr(g[f"epoch{e}_offsets"], g[f"epoch{e}_items"]), f"epoch {e}"


def test_sampler_properties_and_rank_sharding():
    g = load(golden_files("feed_sampler_n400_b8_thr")[0])
    lens, thr = g["seq_len"], int(g["single_sent_threshold"])
    whole = list(_sampler(g))
    seen = sorted(i for b in whole for i in b)
    assert seen == list(range(len(lens)))                                             # a partition of the data set
    for b in whole:
        assert len(b) == 1 or all(lens[i] < thr for i in b)                           # over-long sentences travel alone
        assert all(lens[b[q]] >= lens[b[q + 1]] for q in range(len(b) - 1))           # decreasing length inside a batch
        assert len(b) <= int(g["max_sentence"])
    for world in (2, 3, 4, 7):
        samplers = [_sampler(g, rank=r, world_size=world) for r in range(world)]     # same seed -> same epoch list on every rank
        shards = [list(sm) for sm in samplers]
        # every rank runs the SAME number of steps (a data-parallel step ends in a collective): the epoch's batch list is
        # wrapped around to a multiple of the world size, so the union is the whole epoch plus < world repeated batches
        per = -(-len(whole) // world)
        assert all(len(s) == per for s in shards) and all(len(sm) == per for sm in samplers)
        dealt = [tuple(b) for r in range(per) for s in shards for b in [s[r]]]      # round-robin order = the padded epoch list
        assert dealt[:len(whole)] == [tuple(b) for b in whole]
        assert dealt[len(whole):] == [tuple(b) for b in whole[:per * world - len(whole)]]
    # fewer batches than ranks: the wrap is cyclic, so every rank still gets a batch
    world = 3 * len(whole) + 1
    shards = [list(_sampler(g, rank=r, world_size=world)) for r in range(world)]
    assert all(len(s) == 1 for s in shards)
    assert [tuple(s[0]) for s in shards] == [tuple(whole[i % len(whole)]) for i in range(world)]


@pytest.mark.parametrize("seed", range(6))
def test_kmeans_and_batches_vs_oracle(seed, oracle_mod):
    from vlgae_amd import _C
    rng = np.random.default_rng(seed)
    n = int(rng.integers(20, 3000))
    k = int(rng.integers(2, 20))
    x = np.clip(rng.poisson(rng.integers(2, 20), n) + 1, 1, 60).astype(np.int32)
    d = np.unique(x)
    c0 = rng.permutation(d)[:k].astype(np.float32)
    want_c, want_y = oracle_mod.feed_kmeans(x, c0, k)
    cent, y, m = np.empty(k, np.float32), np.empty(n, np.int32), ctypes.c_int(0)
    pad = np.concatenate([c0, np.full(k - len(c0), np.inf, np.float32)])
    p = lambda a: a.ctypes.data_as(ctypes.c_void_p)
    _C.check(_C.lib().vlg_feed_kmeans(p(x), n, p(pad), k, 32, p(cent), p(y), ctypes.byref(m)), "kmeans")
    assert m.value == len(want_c) and np.array_equal(cent[:m.value], want_c) and np.array_equal(y, want_y)
    # batches from those buckets under random permutations
    buckets = [np.nonzero(y == j)[0].tolist() for j in range(m.value)]
    chunks = [int(rng.integers(1, len(b) + 1)) for b in buckets]
    perms = [rng.permutation(len(b)) for b in buckets]
    order = rng.permutation(sum(chunks))
    thr = int(rng.choice([-1, 12]))
    want = oracle_mod.feed_batches(x, buckets, chunks, perms, order, thr, bool(seed % 2))
    off = np.cumsum([0] + [len(b) for b in buckets]).astype(np.int64)
    items = np.asarray([i for b in buckets for i in b], np.int64)
    ch, pp, oo = np.asarray(chunks, np.int64), np.concatenate(perms).astype(np.int64), order.astype(np.int64)
    o_off, o_items, nb = np.empty(sum(chunks) + n + 1, np.int64), np.empty(n, np.int64), ctypes.c_int64(0)
    _C.check(_C.lib().vlg_feed_batches(p(x), n, p(off), p(items), len(buckets), p(ch), p(pp), p(oo), thr, seed % 2, p(o_off), p(o_items),
                                       ctypes.byref(nb)), "batches")
    assert _uncsr(o_off[:nb.value + 1], o_items) == want


def test_feed_argument_errors():
    from vlgae_amd import _C
    lib = _C.lib()
    x = np.asarray([3, 4, 5], np.int32)
    c = np.asarray([3, 4, 5, 6], np.float32)
    p = lambda a: a.ctypes.data_as(ctypes.c_void_p)
    out_c, out_y, m = np.empty(4, np.float32), np.empty(3, np.int32), ctypes.c_int(0)
    assert lib.vlg_feed_kmeans(p(x), 3, p(c), 4, 32, p(out_c), p(out_y), ctypes.byref(m)) == 0x1003      # k > n
    assert b"k <= n" in lib.vlg_last_error()
    off, items, ch = np.asarray([0, 3], np.int64), np.asarray([0, 1, 2], np.int64), np.asarray([5], np.int64)
    perm, order = np.asarray([0, 1, 2], np.int64), np.arange(5, dtype=np.int64)
    o_off, o_items, nb = np.empty(9, np.int64), np.empty(3, np.int64), ctypes.c_int64(0)
    assert lib.vlg_feed_batches(p(x), 3, p(off), p(items), 1, p(ch), p(perm), p(order), -1, 1, p(o_off), p(o_items), ctypes.byref(nb)) == 0x1003
    ch[0], perm[2] = 2, 7
    assert lib.vlg_feed_batches(p(x), 3, p(off), p(items), 1, p(ch), p(perm), p(order), -1, 1, p(o_off), p(o_items), ctypes.byref(nb)) == 0x1003
    assert b"permutation" in lib.vlg_last_error()


def _write_files(g, root):
    n = len(g["sg_nobj"])
    for i in range(n):
        np.save(os.path.join(root, f"{i}.npy"), g[f"file{i}"])
    sg = {i: {"obj": list(range(int(g["sg_nobj"][i]))), "rel": []} for i in range(n)}
    for i, s, o in g["sg_rel"]:
        sg[int(i)]["rel"].append({"subj": int(s), "obj": int(o)})
    return sg, [(k, {"img_id": int(i)}) for k, i in enumerate(g["order"])]


def _check_collate(g, tag, a, b):
    base = "lead" if tag in ("lead", "gold") else "sample6"
    assert a["vis_box_feat"].dtype == torch.float32 and b["vis_box"].dtype == torch.float32
    assert np.array_equal(a["vis_box_feat"].cpu().numpy(), g[base + "_feat_f16"].astype(np.float32))
    assert np.array_equal(b["vis_box"].cpu().numpy(), g[tag + "_box"])
    assert a["vis_box_mask"].dtype == torch.bool and np.array_equal(a["vis_box_mask"].cpu().numpy(), g[tag + "_mask"])
    assert a["vis_rel_mask"].dtype == torch.bool and np.array_equal(a["vis_rel_mask"].cpu().numpy(), g[tag + "_rel"])
    assert np.array_equal(a["vis_available"].cpu().numpy(), g[tag + "_available"])


@pytest.mark.parametrize("tag,sample,gold", [("lead", 0, False), ("sample6", 6, False), ("gold", 0, True), ("gold_sample6", 6, True)])
@pytest.mark.parametrize("threads", [1, 4])
def test_collate_matches_reference(tmp_path, tag, sample, gold, threads):
    from vlgae_amd.feed import RegionFeatLoader
    g = load(golden_files("feed_collate_")[0])
    sg, batch = _write_files(g, tmp_path)
    np.random.seed(5)
    a, b = RegionFeatLoader(tmp_path, sg, sample, gold, threads=threads)(batch)
    _check_collate(g, tag, a, b)


def test_collate_errors(tmp_path):
    from vlgae_amd.feed import RegionFeatLoader
    np.save(tmp_path / "0.npy", np.zeros((3, 2052), np.float32))
    np.save(tmp_path / "1.npy", np.zeros((3, 100), np.float32))
    np.save(tmp_path / "2.npy", np.zeros((3, 2052), np.int32))
    np.save(tmp_path / "3.npy", np.asfortranarray(np.zeros((3, 2052), np.float32)))
    (tmp_path / "4.npy").write_bytes(b"not an array")
    ld = RegionFeatLoader(tmp_path)
    ld([(0, {"img_id": 0})])
    for bad, text in ((1, "columns"), (2, "dtype"), (3, "fortran"), (4, "not a .npy")):
        with pytest.raises(RuntimeError, match=text):
            ld([(0, {"img_id": 0}), (1, {"img_id": bad})])
    with pytest.raises(AssertionError):
        ld([(0, {"img_id": 99})])


@pytest.mark.gpu
def test_collate_to_device_through_pinned_staging(tmp_path):
    from vlgae_amd.feed import RegionFeatLoader
    g = load(golden_files("feed_collate_")[0])
    sg, batch = _write_files(g, tmp_path)
    ld = RegionFeatLoader(tmp_path, sg, 0, True, device="cuda:0")
    for _ in range(3):   # the side stream is reused; the consumer's stream waits on it
        a, b = ld(batch)
        assert a["vis_box_feat"].is_cuda and b["vis_box"].is_cuda and a["vis_box_mask"].is_cuda
        s = a["vis_box_feat"].sum()   # consumer on the current stream
        _check_collate(g, "gold", a, b)
        assert float(s) == float(g["lead_feat_f16"].astype(np.float64).sum())


@pytest.mark.parametrize("path", [p for p in golden_files("feed_sampler_") if "same_len" not in p],
                         ids=[i for i in golden_ids("feed_sampler_") if "same_len" not in i])
def test_oracle_feed_matches_reference(path, oracle_mod):
    """Pins the oracle's restatement itself on the reference's buckets and first epoch."""
    g = load(path)
    x = g["seq_len"]
    k = min(len(x), int(g["num_bucket"]))
    torch.manual_seed(int(g["torch_seed"]))
    d = torch.from_numpy(x).float().unique()
    c0 = d[torch.randperm(len(d))[:k]].numpy()
    cent, y = oracle_mod.feed_kmeans(x, c0, k)
    assert np.array_equal(cent.astype(np.float64), g["sizes"])
    buckets = [np.nonzero(y == j)[0].tolist() for j in range(len(cent))]
    assert buckets == _uncsr(g["bucket_offsets"], g["bucket_items"])
    gen = torch.Generator().manual_seed(1)
    perms = [torch.randperm(len(b), generator=gen).tolist() for b in buckets]
    order = torch.randperm(int(g["chunks"].sum()), generator=gen).tolist()
    got = oracle_mod.feed_batches(x, buckets, g["chunks"].tolist(), perms, order, int(g["single_sent_threshold"]), bool(g["sort_in_batch"]))
    assert got == _uncsr(g["epoch0_offsets"], g["epoch0_items"])
