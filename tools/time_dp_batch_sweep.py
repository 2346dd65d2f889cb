"""DMV1o inside + outside (Log, bf16 potentials, L = 40) against the batch size: is B = 256 one workgroup per CU (time = the critical path of one
sentence), and what does a second workgroup per CU cost?    python tools/time_dp_batch_sweep.py"""
import sys, torch
sys.path.insert(0, '.')
from vlgae_amd.torch_struct import functional as F
import vlgae_amd.torch_struct as ts
dev = torch.device('cuda:0')
L = 40
def timeit(fn, n=100):
    for _ in range(10): fn()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) * 1e3 / n
g = torch.Generator().manual_seed(1)
for B in (32, 64, 128, 192, 224, 256, 288, 320, 384, 512, 768, 1024, 2048):
    md, ma = ts.DMV1o.merge(torch.randn(B, L, 2, 2, 2, generator=g).log_softmax(-1).to(dev), torch.randn(B, L, L, 2, generator=g).to(dev),
                            torch.randn(B, L, generator=g).log_softmax(-1).to(dev))
    md, ma = md.bfloat16(), ma.bfloat16()
    lengths = torch.full((B,), L, dtype=torch.long, device=dev)
    t = timeit(lambda: F.dmv1o_run(md, ma, lengths, 0, True))
    print(f"B={B}: {t:.1f} us, {B / t:.2f} M sentences/s", flush=True)
