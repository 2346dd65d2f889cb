"""The grounding loss's forward pass alone (arg-max alignment + POS prior + cross-entropies), n calls: the program rocprofv3 is pointed at by
tools/time_argmax_ablation.sh.   python tools/argmax_fwd.py [masked|unmasked] [shipped]"""
import sys, torch
sys.path.insert(0, '.')
from vlgae_amd import align, encoders
dev = torch.device('cuda:0')
masked = 'unmasked' not in sys.argv
shipped = 'shipped' in sys.argv
B, L, R, d = (64, 40, 36, 128) if shipped else (256, 40, 36, 128)
Q = 2 * (L + 1)
g = torch.Generator().manual_seed(0)
lengths = torch.randint(L // 2, L + 1, (B,), generator=g) if masked else torch.full((B,), L)
m1 = torch.cat([torch.zeros(B, 1, dtype=torch.bool), torch.arange(L)[None] < lengths[:, None]], 1)
tmask = torch.cat([m1, m1], 1).to(dev) if masked else torch.ones(B, Q, dtype=torch.bool, device=dev)
n_box = torch.randint((3 * R) // 5, R + 1, (B,), generator=g) if masked else torch.full((B,), R)
box_mask = (torch.arange(R)[None] < n_box[:, None]).to(dev)
vmask = encoders.factor_mask(box_mask, shipped, shipped, shipped)
V = vmask.shape[1]
marg = (torch.rand(B, Q, generator=g).to(dev) * tmask)
txt = (torch.randn(B, Q, d, generator=g) * 0.5).to(dev, torch.bfloat16)
vis = (torch.randn(B, V, d, generator=g) * 0.5).to(dev, torch.bfloat16)
with torch.no_grad():
    for _ in range(30):
        total, sums = align.grounding_loss_factor_ce(txt, vis, tmask, vmask, marg, int(lengths.sum()), 1.0)
torch.cuda.synchronize()
print('ok', float(total))
