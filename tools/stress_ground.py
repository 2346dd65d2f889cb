"""Randomised sweep of the grounding loss (bf16: LDS-strip cross-entropies + dense matrix-core backward, config-2 widths and wide
region axes) against the fp64 oracle, with every case run five times for bit-reproducibility (a race in the W construction or
in the partial sums would show up as differing bits).  Run on the GPU box: python tools/stress_ground.py"""
import sys, numpy as np, torch
sys.path.insert(0, '.')
import oracle
from vlgae_amd import align
oracle.build()
dev = torch.device('cuda:0')
rng = np.random.default_rng(int(sys.argv[1]) if len(sys.argv) > 1 else 2026)
worst = 0.0
for it in range(int(sys.argv[2]) if len(sys.argv) > 2 else 40):
    wide = it % 4 == 3
    B = int(rng.integers(1, 70)) if not wide else int(rng.integers(1, 12))
    L = int(rng.integers(1, 48))
    V = int(rng.integers(1, 65)) if not wide else int(rng.choice([65, 96, 129, 300, 520, 1369]))
    Q, d = 2 * (L + 1), 128
    lengths = rng.integers(max(1, L // 2), L + 1, B)
    m1 = np.concatenate([np.zeros((B, 1), bool), np.arange(L)[None] < lengths[:, None]], 1)
    tmask = np.concatenate([m1, m1], 1)
    vmask = rng.random((B, V)) > rng.choice([0.0, 0.15, 0.5])
    vmask[:, 0] = True
    txt = torch.from_numpy((rng.standard_normal((B, Q, d)) * 0.4).astype(np.float32)).bfloat16()
    vis = torch.from_numpy((rng.standard_normal((B, V, d)) * 0.4).astype(np.float32)).bfloat16()
    marg = (rng.random((B, Q)) * tmask).astype(np.float32)
    num = int(lengths.sum())
    pen = seg = None
    if it % 2 == 1 and V >= 4:   # every other case with the POS prior of the diagonal pairs (joint.py:446-470)
        tag = rng.integers(0, 6, (B, L))
        pen, seg = oracle.grounding_prior(tag, ["obj", "rel", "img"], [V // 2, V - V // 2 - 1, 1],
                                          dict(obj=np.array([0, 1]), rel=np.array([1, 2]), attr=np.array([5])), Q)
        ref = oracle.grounding_loss(txt.float().numpy(), vis.float().numpy(), tmask, vmask, marg, num, 1.0, pen, seg, -1e20, np.float64)
    else:
        ref = oracle.grounding_loss(txt.float().numpy(), vis.float().numpy(), tmask, vmask, marg, num, 1.0, dtype=np.float64)
    tt, tv = txt.to(dev).requires_grad_(True), vis.to(dev).requires_grad_(True)
    tm, vm, mg = (torch.from_numpy(a).to(dev) for a in (tmask, vmask, marg))
    first = None
    for rep in range(5):
        if pen is None:
            total, sums = align.grounding_loss_factor_ce(tt, tv, tm, vm, mg, num, 1.0)
        else:
            total, sums = align.grounding_loss_factor_ce(tt, tv, tm, vm, mg, num, 1.0, torch.from_numpy(pen.astype(np.float32)).to(dev),
                                                         torch.from_numpy(seg).to(dev))
        g_txt, g_vis = torch.autograd.grad(total, [tt, tv])
        cur = (sums.clone(), g_txt.clone(), g_vis.clone())
        if first is None:
            first = cur
        else:
            assert all(torch.equal(a, b) for a, b in zip(first, cur)), ("not reproducible", it, B, L, V, rep)
    s_ = first[0].cpu().numpy()
    e = [abs(s_[0] - ref["txt2vis"]) / max(1.0, abs(ref["txt2vis"])), abs(s_[1] - ref["vis2txt"]) / max(1.0, abs(ref["vis2txt"]))]
    for got, want in ((first[1], ref["g_txt"]), (first[2], ref["g_vis"])):
        e.append(np.abs(got.float().cpu().numpy() - want).max() / max(1.0, np.abs(want).max()))
    assert e[0] <= 1e-4 and e[1] <= 1e-4 and e[2] <= 1e-2 and e[3] <= 1e-2, (it, B, L, V, e)
    worst = max(worst, *e[2:])
print("stress ok; worst gradient error (bf16 outputs) %.2e of max|g|" % worst)
