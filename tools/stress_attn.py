"""Randomised sweep of the attention fuse (joint.py:670-674): forward and all six gradients against the fp64 oracle over odd shapes --
one-pass and key-split launches (automatic and forced chunk sizes), both dtypes, feature widths that are not powers of two, single
words / single keys, batches that do not fill the grid -- with every case run three times for bit-reproducibility.
Run on the GPU box: python tools/stress_attn.py [seed] [cases]"""
import sys, numpy as np, torch
sys.path.insert(0, '.')
import oracle
from vlgae_amd import _C, align
oracle.build()
dev = torch.device('cuda:0')
rng = np.random.default_rng(int(sys.argv[1]) if len(sys.argv) > 1 else 2026)
t = lambda a: torch.from_numpy(np.ascontiguousarray(a)).to(dev)
worst = [0.0, 0.0]
n_split = 0
for it in range(int(sys.argv[2]) if len(sys.argv) > 2 else 60):
    kind = it % 6
    B = int(rng.integers(1, 9))
    L = int(rng.choice([1, 2, 15, 16, 17, 31, 40, 47, 48, 49, 63]))
    V = int(rng.choice([1, 2, 15, 16, 36, 63, 64])) if kind < 2 else int(rng.choice([65, 66, 127, 128, 129, 255, 257, 300, 511, 640, 1369]))
    d = int(rng.choice([16, 32, 48, 64, 128, 144, 256]))         # (the adjoint takes d, h multiples of 16 up to 256)
    h = int(rng.choice([16, 48, 64, 80, 128, 176, 240, 256]))
    ck = 0 if kind in (0, 1, 2) else int(rng.choice([64, 128, 192, 256]))
    bf16 = it % 2 == 1
    vis, txt = rng.standard_normal((B, V, d)).astype(np.float32) * 0.4, rng.standard_normal((B, L + 1, d)).astype(np.float32) * 0.4
    mid, enc = rng.standard_normal((B, V, h)).astype(np.float32), rng.standard_normal((B, L, h)).astype(np.float32)
    gm, bt = (rng.random(h) + 0.5).astype(np.float32), rng.standard_normal(h).astype(np.float32)
    dout = rng.standard_normal((B, L, h)).astype(np.float32)
    arrs = [vis, txt, mid, enc]
    if bf16:
        arrs = [torch.from_numpy(a).bfloat16().float().numpy() for a in arrs]
    _, ref_out = oracle.attn_fuse(*arrs, gm, bt, 1e-5, np.float64)
    ref = oracle.attn_fuse_backward(*arrs, gm, dout, 1e-5, np.float64)
    leaves = [t(a) for a in arrs]
    if bf16:
        leaves = [a.bfloat16() for a in leaves]
    leaves += [t(gm), t(bt)]
    for a in leaves:
        a.requires_grad_(True)
    n_split += int(_C.lib().vlg_attn_fuse_workspace(B, L, V, h, ck) > 0)
    first = None
    for rep in range(3):
        out = align.attention_fuse(*leaves, 1e-5, key_chunk=ck)
        grads = torch.autograd.grad(out, leaves, t(dout))
        cur = [out.detach().clone()] + [g.clone() for g in grads]
        if first is None:
            first = cur
        else:
            assert all(torch.equal(a, b) for a, b in zip(first, cur)), ("not reproducible", it, B, L, V, d, h, ck, bf16, rep)
    case = (it, B, L, V, d, h, ck, "bf16" if bf16 else "f32")
    e_out = np.abs(first[0].cpu().numpy() - ref_out).max()
    assert e_out <= 1e-4, (case, "out", e_out)
    for i, (got, want) in enumerate(zip(first[1:], ref)):
        low = bf16 and i < 4                                     # bf16 gradients are rounded once on return
        e = np.abs(got.float().cpu().numpy() - want).max() / max(1.0, np.abs(want).max())
        tol = 1e-2 if low else 1e-4
        assert e <= tol, (case, i, e)
        worst[0 if low else 1] = max(worst[0 if low else 1], e)
    assert not first[2][:, 0].any(), (case, "root slot")
print("stress ok: %d key-split cases; worst gradient error %.2e (bf16 outputs) / %.2e (fp32 outputs) of max|g|" % (n_split, worst[0], worst[1]))
