import sys, numpy as np, torch
sys.path.insert(0, '.')
import oracle
from vlgae_amd.torch_struct import functional as F
dev = torch.device('cuda:0')
def t(a): return torch.from_numpy(np.ascontiguousarray(a)).to(dev)
B, L, seed = 4, 120, 14
rng = np.random.default_rng(seed)
dec = rng.standard_normal((B, L, 2, 2, 2)).astype(np.float32)
dec = dec - np.log(np.exp(dec).sum(-1, keepdims=True))
attach = rng.standard_normal((B, L, L, 2)).astype(np.float32)
root = rng.standard_normal((B, L)).astype(np.float32)
lengths = rng.integers(1, L + 1, size=B); lengths[0] = L
md, ma = oracle.dmv1o_merge(dec, attach, root)
ref = {n: oracle.dmv1o(md, ma, lengths, n, np.float64) for n in ('log', 'max')}
def run(variant):
    bad = 0
    for it in range(20):
        for sr, name in ((0, 'log'), (1, 'max')):
            if variant == 'temps':
                lz, gd, ga = F.dmv1o_run(t(md), t(ma), t(lengths), sr, True)
                lz0, _, _ = F.dmv1o_run(t(md), t(ma), t(lengths), sr, False)
            elif variant == 'temps_sync':
                lz, gd, ga = F.dmv1o_run(t(md), t(ma), t(lengths), sr, True); torch.cuda.synchronize()
                lz0, _, _ = F.dmv1o_run(t(md), t(ma), t(lengths), sr, False); torch.cuda.synchronize()
            elif variant == 'persist':
                lz, gd, ga = F.dmv1o_run(tmd, tma, tl, sr, True)
                lz0, _, _ = F.dmv1o_run(tmd, tma, tl, sr, False)
            e = np.abs(lz.cpu().numpy() - ref[name][0][:, 0]).max()
            e0 = np.abs(lz0.cpu().numpy() - ref[name][0][:, 0]).max()
            eg = np.abs(ga.cpu().numpy() - ref[name][2]).max()
            if e > 1e-3 or e0 > 1e-3 or eg > 1e-3:
                bad += 1
                if bad < 4: print(variant, it, name, 'fused', e, 'inside', e0, 'grad', eg)
    print(variant, 'bad', bad, 'of 40')
tmd, tma, tl = t(md), t(ma), t(lengths)
for v in ('persist', 'temps_sync', 'temps', 'persist'):
    run(v)
