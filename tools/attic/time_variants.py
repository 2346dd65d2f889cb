import sys, time, torch, numpy as np
sys.path.insert(0, '.')
from vlgae_amd import _C
from vlgae_amd.torch_struct import functional as F
import vlgae_amd.torch_struct as ts
dev = torch.device('cuda:0')
def synth(B, L):
    g = torch.Generator().manual_seed(1)
    dec = torch.randn(B, L, 2, 2, 2, generator=g).log_softmax(-1).to(dev)
    attach = torch.randn(B, L, L, 2, generator=g).to(dev)
    root = torch.randn(B, L, generator=g).log_softmax(-1).to(dev)
    return ts.DMV1o.merge(dec, attach, root)
def timeit(fn, n=50):
    for _ in range(5): fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) * 1e3 / n
for B, L in ((256, 40), (256, 20), (256, 10), (1024, 40), (4096, 40), (256, 80)):
    md, ma = synth(B, L)
    lengths = torch.full((B,), L, dtype=torch.long, device=dev)
    res = {}
    for sr, name in ((0, 'log'), (1, 'max')):
        res[name + '_io'] = timeit(lambda: F.dmv1o_run(md, ma, lengths, sr, True))
        res[name + '_in'] = timeit(lambda: F.dmv1o_run(md, ma, lengths, sr, False))
    arc = ma[..., 0].contiguous()
    res['dep_io'] = timeit(lambda: F.deptree_run(arc, lengths, 0, True))
    print(f'B={B} L={L}: ' + ' '.join(f'{k}={v:.1f}us' for k, v in res.items()), flush=True)
