"""Host-side data feed timings (row f4): bucketing + one epoch of batches at MSCOCO scale, and the region-feature collate of
one config-2 batch (256 images x 35 regions x (2048 + 4) float32 = 73.5 MB) from .npy files in the page cache."""
import os, sys, tempfile, time
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from vlgae_amd.feed import ConstantTokenNumSampler, RegionFeatLoader

rng = np.random.default_rng(0)
n = 400_000
lens = np.clip(rng.poisson(10, n) + 2, 1, 50).tolist()
torch.manual_seed(0)
t = time.perf_counter(); sm = ConstantTokenNumSampler(lens, 5000, -1, 16); t_init = time.perf_counter() - t
t = time.perf_counter(); batches = list(sm); t_epoch = time.perf_counter() - t
print("sampler n=%d: buckets + first epoch %.3f s, next epoch %.3f s, %d batches" % (n, t_init, t_epoch, len(batches)))
dev = "cuda:0" if torch.cuda.is_available() else None
with tempfile.TemporaryDirectory() as td:
    B = 256
    for i in range(B):
        np.save(os.path.join(td, f"{i}.npy"), rng.standard_normal((36 + i % 20, 2052)).astype(np.float32))
    batch = [(i, {"img_id": i}) for i in range(B)]
    for threads in (1, 4, 8, 16):
        ld = RegionFeatLoader(td, threads=threads, device=dev)
        ld(batch)
        t = time.perf_counter()
        for _ in range(5):
            out = ld(batch)
        if dev: torch.cuda.synchronize()
        dt = (time.perf_counter() - t) / 5
        print("collate B=%d threads=%d device=%s: %.2f ms  (%.2f GB/s of features)" % (B, threads, dev, dt * 1e3, B * 35 * 2052 * 4 / dt / 1e9))
