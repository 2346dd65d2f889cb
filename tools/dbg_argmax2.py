import os, sys, numpy as np, torch
sys.path.insert(0, '.')
from vlgae_amd import align
dev = torch.device('cuda:0')
rng = np.random.default_rng(0)
B, L, V, d = 6, 40, 36, 128
Q = 2 * (L + 1)
t = lambda a: torch.from_numpy(np.ascontiguousarray(a)).to(dev)
txt = t((rng.standard_normal((B, Q, d)) * 0.5).astype(np.float32)).bfloat16()
vis = t((rng.standard_normal((B, V, d)) * 0.5).astype(np.float32)).bfloat16()
marg = t(rng.random((B, Q)).astype(np.float32))
pen = t((rng.integers(0, 3, (B, Q, 3)) * 100.0).astype(np.float32)); seg = t(rng.integers(0, 3, V).astype(np.uint8))
tm1 = torch.ones(B, Q, dtype=torch.bool, device=dev); vm1 = torch.ones(B, V, dtype=torch.bool, device=dev)
def go(tm, vm):
    with torch.no_grad():
        return align.grounding_loss_factor_ce(txt, vis, tm, vm, marg, 200, 1.0, pen, seg)[1].cpu().numpy()
for name, tm, vm in (('none', None, None), ('tmask only', tm1, None), ('vmask only', None, vm1), ('both', tm1, vm1)):
    os.environ["VLG_ALIGN_ARGMAX_OLD"] = "1"
    o = go(tm, vm)
    os.environ.pop("VLG_ALIGN_ARGMAX_OLD")
    n = go(tm, vm)
    print('%-12s old %s new %s' % (name, o, n))
S = torch.einsum('bqd,bvd->bqv', txt.float(), vis.float())
Sd = S - pen[:, :, seg.long()]
print('diag maxV sum %.4f  maxQ sum %.4f' % (Sd.max(2).values.sum().item(), Sd.max(1).values.sum().item()))
