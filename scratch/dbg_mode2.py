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
print('lengths', lengths)
md, ma = oracle.dmv1o_merge(dec, attach, root)
tmd, tma, tl = t(md), t(ma), t(lengths)
for sr, name in ((0, 'log'), (1, 'max'), (1, 'max'), (0, 'log')):
    ref_lz, ref_gd, ref_ga = oracle.dmv1o(md, ma, lengths, name, np.float64)
    lz, gd, ga = F.dmv1o_run(tmd, tma, tl, sr, True)
    torch.cuda.synchronize()
    lz0, _, _ = F.dmv1o_run(tmd, tma, tl, sr, False)
    torch.cuda.synchronize()
    print(name, 'fused', lz.cpu().numpy(), 'inside', lz0.cpu().numpy(), 'ref', ref_lz[:, 0])
    print('   grad err', np.abs(ga.cpu().numpy() - ref_ga).max(), np.abs(gd.cpu().numpy() - ref_gd).max())
