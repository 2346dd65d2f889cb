import sys, numpy as np, torch
sys.path.insert(0, '.')
from vlgae_amd import align
dev = torch.device('cuda:0')
rng = np.random.default_rng(5)
B, A, Q, V, d = 3, 5, 7, 3, 32
txt = torch.randint(-3, 4, (B, Q, d), device=dev).float()
vis = torch.randint(-3, 4, (A, V, d), device=dev).float()
tm = torch.from_numpy(rng.random((B, Q)) > 0.2).to(dev)
vm = torch.from_numpy(rng.random((A, V)) > 0.2).to(dev)
ones_t, ones_v = torch.ones_like(tm), torch.ones_like(vm)
base = torch.einsum('avd,bqd->baqv', vis, txt)
for name, t_, v_ in (('all-true', ones_t, ones_v), ('tmask only', tm, None), ('vmask only', None, vm), ('both', tm, vm)):
    ref = base.clone()
    if v_ is not None: ref = ref.masked_fill(~v_[None, :, None, :], -1e20)
    if t_ is not None: ref = ref.masked_fill(~t_[:, None, :, None], -1e20)
    out = align.bilinear_align(txt, vis, t_, v_)['full']
    bad = ((out - ref).abs() > 1e-3)
    print(name, 'n bad', int(bad.sum()))
    if bad.any():
        print('  kernel-masked pattern b=0,a=0:\n', (out[0, 0] < -1e19).int().tolist(), '\n  expected:\n', (ref[0, 0] < -1e19).int().tolist())
print('tm[0]', tm[0].int().tolist(), 'vm[0]', vm[0].int().tolist())
