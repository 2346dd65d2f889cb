"""Randomised DP parity sweep against the fp64 oracle (ragged lengths, N across the placement-mode boundaries, both
semirings, decode and one-hot argmax).  Run on the GPU box: python tools/stress_dp.py"""
import sys, numpy as np, torch
sys.path.insert(0, '.')
import oracle, vlgae_amd.torch_struct as ts
from vlgae_amd.torch_struct import functional as F
oracle.build()
dev = torch.device('cuda:0')
rng = np.random.default_rng(int(sys.argv[1]) if len(sys.argv) > 1 else 123)
worst = 0.0
for it in range(int(sys.argv[2]) if len(sys.argv) > 2 else 36):
    B = int(rng.integers(1, 9)); L = int(rng.choice([1, 2, 3, 5, 9, 17, 31, 40, 59, 60, 61, 62, 63, 75, 80, 88, 89, 100]))
    lengths = rng.integers(1, L + 1, B); lengths[0] = L
    dec = np.log(rng.dirichlet(np.ones(2), (B, L, 2, 2))).astype(np.float32)
    attach = (rng.standard_normal((B, L, L, 2)) * rng.choice([0.5, 2.0, 6.0])).astype(np.float32)
    root = np.log(rng.dirichlet(np.ones(L), B)).astype(np.float32)
    md, ma = ts.DMV1o.merge(torch.from_numpy(dec).to(dev), torch.from_numpy(attach).to(dev), torch.from_numpy(root).to(dev))
    ln = torch.from_numpy(lengths).to(dev)
    mdn, man = md.cpu().numpy(), ma.cpu().numpy()
    for sr in (0, 1):
        ref = oracle.dmv1o(mdn, man, lengths, semiring=('log' if sr == 0 else 'max'), grad=True, dtype=np.float64)
        logZ, gdec, gatt = F.dmv1o_run(md, ma, ln, sr, True)
        if ref is None: break
        rz, rgd, rga = ref[0][:, 0], ref[1], ref[2]
        ez = np.abs(logZ.cpu().numpy() - rz).max() / max(1.0, np.abs(rz).max())
        if sr == 0:
            eg = max(np.abs(gdec.cpu().numpy() - rgd).max(), np.abs(gatt.cpu().numpy() - rga).max())
        else:   # Max semiring: the counts are the 0 / 1 indicators of ONE tree (round 4: from the back-pointer walk, every placement mode).
            # Chart values are exact in either precision order (max of two-term fp32 sums), so the fp32 oracle picks the same tree.
            r32m = oracle.dmv1o(mdn, man, lengths, semiring='max', grad=True, dtype=np.float32)
            assert np.array_equal(gdec.cpu().numpy(), r32m[1]) and np.array_equal(gatt.cpu().numpy(), r32m[2]), (it, B, L, 'max counts')
            bv, vd, va, vh = F.dmv1o_viterbi(md, ma, ln)
            assert np.array_equal(vd.cpu().numpy(), r32m[1]) and np.array_equal(va.cpu().numpy(), r32m[2]), (it, B, L, 'viterbi counts')
            eg = 0.0
        worst = max(worst, ez, eg)
        if not (ez < 3e-5 and eg < 1e-4):   # long sentences with peaky scores: is it fp32 rounding?  ask the fp32 oracle
            r32 = oracle.dmv1o(mdn, man, lengths, semiring=('log' if sr == 0 else 'max'), grad=True, dtype=np.float32)
            e32 = max(np.abs(r32[1] - rgd).max(), np.abs(r32[2] - rga).max()) if sr == 0 else 0.0
            print(f'  note: it={it} B={B} L={L} sr={sr}: GPU err {eg:.2e}, fp32 CPU oracle err {e32:.2e} vs fp64')
            # fp32 charts over 2(N-1) widths: on long sentences with peaky scores (|score| ~ 20) the GPU's butterfly order and
            # 1-ulp exp2 / log2 land within a small multiple of what a sequential fp32 evaluation gives (observed up to 5.3x at
            # L = 89); the parity target (1e-4 at L = 40) is checked by the tests, this sweep guards against gross regressions
            assert ez < 3e-5 and eg < max(1e-4, 8 * e32), (it, B, L, sr, ez, eg, e32)
    # decode: heads give a projective tree whose score equals the Max-semiring value
    best, heads = F.dmv1o_decode(md, ma, ln)
    mx = F.dmv1o_run(md, ma, ln, 1, False)[0]
    assert torch.allclose(best, mx)
    h = heads.cpu().numpy()
    for b in range(B):
        assert oracle.is_projective_tree(h[b], int(lengths[b])), (it, b)
        sc = oracle.dmv1o_tree_score(mdn[b], man[b], h[b], int(lengths[b]))
        assert abs(sc - float(mx[b])) <= 2e-4 * max(1.0, abs(sc)), (it, b, sc, float(mx[b]))
    # one-hot argmax equals heads
    am = ts.DMV1o([md, ma], ln).argmax.sum(-1)
    for b in range(B):
        nz = am[b].nonzero().cpu().numpy()
        assert len(nz) == lengths[b] and all(h[b][c] == hh for hh, c in nz), (it, b)
print('stress ok; worst rel err', worst)
