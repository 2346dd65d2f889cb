"""A-B of align_argmax_kernel vs align_max_kernel<true,3> inside the grounding loss (sums only), small masked shapes."""
import os, sys, numpy as np, torch
sys.path.insert(0, '.')
from vlgae_amd import align
dev = torch.device('cuda:0')
def run(B, L, V, with_pen, masked, seed=0):
    rng = np.random.default_rng(seed)
    Q = 2 * (L + 1); d = 128
    lengths = rng.integers(max(1, L // 2), L + 1, B)
    m1 = np.concatenate([np.zeros((B, 1), bool), np.arange(L)[None] < lengths[:, None]], 1)
    tmask = np.concatenate([m1, m1], 1) if masked else np.ones((B, Q), bool)
    vmask = rng.random((B, V)) > 0.2 if masked else np.ones((B, V), bool)
    vmask[:, 0] = True
    t = lambda a: torch.from_numpy(np.ascontiguousarray(a)).to(dev)
    txt = t((rng.standard_normal((B, Q, d)) * 0.5).astype(np.float32)).bfloat16()
    vis = t((rng.standard_normal((B, V, d)) * 0.5).astype(np.float32)).bfloat16()
    marg = t((rng.random((B, Q)) * tmask).astype(np.float32))
    pen = seg = None
    if with_pen:
        pen = t((rng.integers(0, 3, (B, Q, 3)) * 100.0).astype(np.float32)); seg = t(rng.integers(0, 3, V).astype(np.uint8))
    out = []
    for old in ("1", ""):
        if old: os.environ["VLG_ALIGN_ARGMAX_OLD"] = "1"
        else: os.environ.pop("VLG_ALIGN_ARGMAX_OLD", None)
        with torch.no_grad():
            total, sums = align.grounding_loss_factor_ce(txt, vis, t(tmask), t(vmask), marg, int(lengths.sum()), 1.0, pen, seg)
        out.append(sums.cpu().numpy())
    print(B, L, V, 'pen' if with_pen else '-', 'masked' if masked else '-', out[0], out[1], 'OK' if np.allclose(out[0], out[1], rtol=1e-5) else 'DIFF')
for B, L, V in ((6, 40, 36), (8, 40, 36), (16, 20, 36), (6, 40, 32)):
    for wp in (False, True):
        for mk in (False, True):
            run(B, L, V, wp, mk)
