"""Data feed for the hot path: token-budget batching of length-bucketed sentences and the
region-feature collate, with the bookkeeping in the native library (csrc/vlg_feed.cpp) and the batch landing in pinned
memory in the padded layout the kernels read -- one DMA per batch, issued on a side stream.

  ConstantTokenNumSampler   mirrors src/datamodule/sampler.py:15-146 (same constructor, same batches for the same torch seed)
  RegionFeatLoader          mirrors _COCODetFeatLazyLoader, src/datamodule/task/vlparse.py:29-114 (same call, same outputs)

Both draw random numbers exactly where the reference does (torch.randperm for the k-means seeds and the epoch's
permutations, np.random.choice for the region sample), so a run that swaps these classes in sees identical batches.
Additions the reference does not have (it is not distributed-aware): `rank` / `world_size` on the
sampler (each rank takes every world_size-th batch of the epoch's list) and `device=` on the loader (pinned staging +
asynchronous copy).
"""
import ctypes
import os
from math import ceil

import numpy as np
import torch

from . import _C


def _i32(a):
    return np.ascontiguousarray(a, dtype=np.int32)


def _i64(a):
    return np.ascontiguousarray(a, dtype=np.int64)


def _p(a):
    return a.ctypes.data_as(ctypes.c_void_p)


def kmeans(x, k, max_it=32):
    """Buckets of `x` (sentence lengths) -> (centroids, clusters) as ConstantTokenNumSampler.kmeans returns them
    (sampler.py:148-191).  The initial centroids are k distinct lengths drawn with the global torch generator."""
    x = _i32(x)
    n, k = len(x), min(len(x), k)
    d = torch.from_numpy(x).float().unique()
    c0 = d[torch.randperm(len(d))[:k]].numpy()   # the reference's draw (sampler.py:157): same generator state, same seeds
    c0 = np.concatenate([c0, np.full(k - len(c0), np.inf, np.float32)])   # fewer distinct lengths than clusters: the rest start empty
    cent = np.empty(k, np.float32)
    y = np.empty(n, np.int32)
    m = ctypes.c_int(0)
    _C.check(_C.lib().vlg_feed_kmeans(_p(x), n, _p(np.ascontiguousarray(c0, np.float32)), k, max_it, _p(cent), _p(y),
                                      ctypes.byref(m)), "feed.kmeans")
    order = np.argsort(y, kind="stable")
    bounds = np.searchsorted(y[order], np.arange(m.value + 1))
    return cent[:m.value].tolist(), [order[bounds[j]:bounds[j + 1]].tolist() for j in range(m.value)]


class ConstantTokenNumSampler:
    """Batch sampler: every batch holds about `max_token` tokens of sentences of similar length."""

    def __init__(self, seq_len, max_token=4096, max_sentence=-1, num_bucket=16, single_sent_threshold=-1, sort_in_batch=True,
                 shuffle=True, force_same_len=False, rank=0, world_size=1):
        assert len(seq_len) >= num_bucket, "The number of samples should be larger than buckets."
        assert num_bucket > 1 or force_same_len, "Use RandomSampler if you do not need bucket."
        self.seq_len = seq_len
        self.max_token = max_token
        self.max_sentence = max_sentence if max_sentence > 0 else 10000000000000000
        self.single_sent_threshold = single_sent_threshold
        self.sort_in_batch = sort_in_batch and not force_same_len
        self.shuffle = shuffle
        self.epoch = 0
        self.rank, self.world_size = rank, world_size
        self._len = _i32(seq_len)
        if force_same_len:   # one bucket per distinct length, in set order like the reference (sampler.py:56-60)
            self.sizes = list(set(seq_len))
            where = {l: i for i, l in enumerate(self.sizes)}
            self.buckets = [[] for _ in self.sizes]
            for i, l in enumerate(seq_len):
                self.buckets[where[l]].append(i)
        else:
            self.sizes, self.buckets = kmeans(seq_len, num_bucket)
        self.chunks = [min(len(b), max(ceil(s * len(b) / max_token), ceil(len(b) / self.max_sentence)))
                       for s, b in zip(self.sizes, self.buckets)]
        self._offsets = _i64(np.cumsum([0] + [len(b) for b in self.buckets]))
        self._items = _i64([i for b in self.buckets for i in b])
        self._chunks = _i64(self.chunks)
        self._batches, self._all_batches = [], []
        self._exhausted = True
        self._init_iter_with_retry()

    def __iter__(self):
        self._init_iter_with_retry()
        yield from self._batches
        self._exhausted = True

    def __len__(self):
        return len(self._batches)

    def _init_iter(self):
        if self.shuffle:
            self.epoch += 1
            g = torch.Generator()
            g.manual_seed(self.epoch)
            draw = lambda m: torch.randperm(m, generator=g).numpy()
        else:
            draw = lambda m: np.arange(m, dtype=np.int64)
        n = len(self._len)
        perms = _i64(np.concatenate([draw(len(b)) for b in self.buckets]))   # drawn bucket by bucket, then the batch order
        n_raw = int(self._chunks.sum())
        order = _i64(draw(n_raw))
        offs = np.empty(n_raw + n + 1, np.int64)
        items = np.empty(n, np.int64)
        nb = ctypes.c_int64(0)
        _C.check(_C.lib().vlg_feed_batches(_p(self._len), n, _p(self._offsets), _p(self._items), len(self.buckets), _p(self._chunks),
                                           _p(perms), _p(order), int(self.single_sent_threshold), int(bool(self.sort_in_batch)),
                                           _p(offs), _p(items), ctypes.byref(nb)), "feed.batches")
        offs, items = offs[:nb.value + 1].tolist(), items.tolist()
        self._all_batches = [items[offs[j]:offs[j + 1]] for j in range(nb.value)]
        if self.world_size > 1:
            # every rank must run the SAME number of steps: the data-parallel step ends in a collective, and a rank with one
            # batch more would wait in it forever.  The epoch's batch list is wrapped around to a multiple of the world size
            # before it is dealt out (rank r takes batches r, r + W, ...), so ranks differ by repeated batches, never by count.
            n_all = len(self._all_batches)
            per = -(-n_all // self.world_size) if n_all else 0
            padded = [self._all_batches[i % n_all] for i in range(per * self.world_size)]   # cyclic: also when n_all < world_size
            self._batches = padded[self.rank::self.world_size]
        else:
            self._batches = self._all_batches
        self._exhausted = False

    def _init_iter_with_retry(self, max_try=5):
        for _ in range(max_try - 1):
            if not self._exhausted:
                return
            self._init_iter()
        if self._exhausted:
            raise ValueError("Failed to init iteration.")

    def set_epoch(self, epoch):
        self.epoch = epoch


class RegionFeatLoader:
    """Collate of the pre-extracted region features: batch of (index, instance) -> the reference's two dicts
    ({vis_box_feat, vis_box_mask, vis_rel_mask, vis_available}, {vis_box}).  `root`/<img_id>.npy holds [regions, 2048 + 4]."""

    FEAT_DIM, BOX_DIM, MAX_REGIONS = 2048, 4, 35

    def __init__(self, root, sg_data=None, sample=0, gold=False, device=None, threads=8):
        self.root = str(root)
        self.sg_data, self.sample, self.gold = sg_data, sample, gold
        self.device = None if device is None else torch.device(device)
        self.threads = threads
        self._stream = None

    def __call__(self, batch):
        lib = _C.lib()
        n = len(batch)
        paths = [os.path.join(self.root, f"{inst['img_id']}.npy").encode() for _, inst in batch]
        n_sel = np.empty(n, np.int32)
        sel = np.zeros((n, max(self.sample, 1)), np.int32) if self.sample > 0 else None
        ids = []
        rows, cols = ctypes.c_int64(0), ctypes.c_int64(0)
        for i, p in enumerate(paths):
            assert os.path.exists(p), p   # the reference asserts on a missing file too (vlparse.py:67)
            _C.check(lib.vlg_feed_npy_shape(p, ctypes.byref(rows), ctypes.byref(cols)), "feed.npy_shape")
            if 0 < self.sample < rows.value:   # vlparse.py:43-47: the same draw from numpy's global generator
                pick = np.random.choice(np.arange(rows.value), self.sample, False)
                sel[i, :self.sample] = pick
            else:
                pick = np.arange(min(rows.value, self.MAX_REGIONS))
                if sel is not None:
                    sel[i, :len(pick)] = pick
            n_sel[i] = len(pick)
            ids.append(pick)
        max_len = int(n_sel.max()) if n else 0
        pin = self.device is not None and self.device.type == "cuda"
        feat = torch.empty(n, max_len, self.FEAT_DIM, pin_memory=pin)
        box = torch.empty(n, max_len, self.BOX_DIM, pin_memory=pin)
        mask = torch.empty(n, max_len, dtype=torch.bool, pin_memory=pin)
        arr = (ctypes.c_char_p * n)(*paths)
        _C.check(lib.vlg_feed_collate_npy(arr, n, None if sel is None else _p(sel), 0 if sel is None else sel.shape[1], _p(n_sel),
                                          self.FEAT_DIM, self.BOX_DIM, max_len, ctypes.c_void_p(feat.data_ptr()),
                                          ctypes.c_void_p(box.data_ptr()), ctypes.c_void_p(mask.data_ptr()), self.threads),
                 "feed.collate_npy")
        rel = None
        if n:
            rel = torch.zeros(n, max_len, max_len, dtype=torch.bool)
            if self.gold:
                for i, (_, inst) in enumerate(batch):
                    m, rm = self.build_gold_mask(inst, ids[i])
                    mask[i] = False
                    mask[i, :len(m)] = m
                    rel[i, :rm.shape[0], :rm.shape[1]] = rm
        if pin:   # one asynchronous copy per tensor on a side stream; the consumer's stream waits on it
            if self._stream is None:
                self._stream = torch.cuda.Stream(self.device)
            cur = torch.cuda.current_stream(self.device)
            # The device tensors are allocated HERE, on the consumer's stream (they belong to ITS allocator pool: when the
            # training loop drops a batch the block is reused in that stream's order), the side stream first waits for
            # the consumer (whatever used the block before has finished) and the consumer joins the copies before the
            # tensors are handed out -- no record_stream bookkeeping (not HIP-graph-capture-safe), no cross-pool reuse.
            host = [feat, box, mask] + ([] if rel is None else [rel])
            dev_t = [torch.empty(t.shape, dtype=t.dtype, device=self.device) for t in host]
            if rel is not None and not rel.is_pinned():
                host[3] = rel.pin_memory()
            self._stream.wait_stream(cur)
            with torch.cuda.stream(self._stream):
                for d, h in zip(dev_t, host):
                    d.copy_(h, non_blocking=True)
            cur.wait_stream(self._stream)
            self._staging = host   # the pinned sources stay referenced until the next call: the copies may still be in flight
            feat, box, mask = dev_t[:3]
            rel = dev_t[3] if rel is not None else None
        return ({"vis_box_feat": feat, "vis_box_mask": mask, "vis_rel_mask": rel, "vis_available": mask[:, 0]},
                {"vis_box": box})

    def build_gold_mask(self, inst, sample_id):
        """Gold scene graph -> (object mask, relation mask over the kept regions), vlparse.py:94-109."""
        sg = self.sg_data[inst["img_id"]]
        n_obj = len(sg["obj"])
        if n_obj == 0:
            return torch.zeros(0, dtype=torch.bool), torch.zeros(0, 0, dtype=torch.bool)
        full = np.zeros((n_obj, n_obj), bool)
        for item in sg["rel"]:
            full[item["subj"], item["obj"]] = True
        pick = np.asarray(sample_id)
        return torch.ones(min(len(pick), n_obj), dtype=torch.bool), torch.from_numpy(full[np.ix_(pick, pick)])
