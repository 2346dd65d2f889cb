import sys, numpy as np, torch
sys.path.insert(0, '.')
from vlgae_amd import align
dev = torch.device('cuda:0')
torch.manual_seed(0)
for dt in (torch.float32, torch.bfloat16):
    B, A, Q, V, d = 2, 3, 20, 18, 64
    # integer-valued data: exact in bf16 and fp32
    txt = torch.randint(-3, 4, (B, Q, d), device=dev).to(dt)
    vis = torch.randint(-3, 4, (A, V, d), device=dev).to(dt)
    out = align.bilinear_align(txt, vis)['full']
    ref = torch.einsum('avd,bqd->baqv', vis.float(), txt.float())
    err = (out - ref).abs()
    print(dt, 'max err', err.max().item(), 'frac wrong', (err > 1e-3).float().mean().item())
    if err.max() > 1e-3:
        bad = (err > 1e-3).nonzero()[:10]
        print(bad.tolist())
        print('out[0,0,:4,:6]', out[0,0,:4,:6].tolist()); print('ref[0,0,:4,:6]', ref[0,0,:4,:6].tolist())
        # is it a transpose / permutation? check if out[0,0] matches ref with k halves swapped etc
        print('row-wise matches:', [(q, [int((out[0,0,q]-ref[0,0,q2]).abs().max() < 1e-3) for q2 in range(Q)].index(1) if any((out[0,0,q]-ref[0,0,q2]).abs().max() < 1e-3 for q2 in range(Q)) else -1) for q in range(8)])
print('--- masks')
rng = np.random.default_rng(5)
for (B, A, Q, V, d) in [(3, 5, 7, 3, 32), (4, 4, 22, 10, 128), (3, 2, 5, 70, 16), (2, 9, 130, 3, 40), (5, 5, 82, 36, 128)]:
    txt = torch.randint(-3, 4, (B, Q, d), device=dev).float()
    vis = torch.randint(-3, 4, (A, V, d), device=dev).float()
    tm = torch.from_numpy(rng.random((B, Q)) > 0.2).to(dev)
    vm = torch.from_numpy(rng.random((A, V)) > 0.2).to(dev)
    ref = torch.einsum('avd,bqd->baqv', vis, txt)
    ref = ref.masked_fill(~vm[None, :, None, :], -1e20).masked_fill(~tm[:, None, :, None], -1e20)
    for dt in (torch.float32, torch.bfloat16):
        r = align.bilinear_align(txt.to(dt), vis.to(dt), tm, vm, max_v=True, max_q=True, diag=(A == B))
        e_full = (r['full'] - ref).abs().max().item()
        e_mv = (r['max_v'] - ref.max(-1).values).abs().max().item()
        e_mq = (r['max_q'] - ref.max(-2).values).abs().max().item()
        print((B, A, Q, V, d), dt, 'full', e_full, 'maxV', e_mv, 'maxQ', e_mq)
        if e_full > 1e-3:
            bad = ((r['full'] - ref).abs() > 1e-3).nonzero()
            print('   n bad', len(bad), bad[:6].tolist(), r['full'][tuple(bad[0].tolist())].item(), ref[tuple(bad[0].tolist())].item())
