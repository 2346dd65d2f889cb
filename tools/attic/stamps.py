import sys, torch
sys.path.insert(0, '.')
from vlgae_amd.torch_struct import functional as F
import vlgae_amd.torch_struct as ts
dev = torch.device('cuda:0')
g = torch.Generator().manual_seed(1)
B, L = 256, 39   # lengths 39 with N = 41 leaves the last row of grad_dec free... use L=40 tensors but lengths 36
L = 40
dec = torch.randn(B, L, 2, 2, 2, generator=g).log_softmax(-1).to(dev); attach = torch.randn(B, L, L, 2, generator=g).to(dev); root = torch.randn(B, L, generator=g).log_softmax(-1).to(dev)
md, ma = ts.DMV1o.merge(dec, attach, root)
md, ma = md.bfloat16().contiguous(), ma.bfloat16().contiguous()
for ln in (32,):
    lengths = torch.full((B,), ln, dtype=torch.long, device=dev)
    for _ in range(3): lz, gd, ga = F.dmv1o_run(md, ma, lengths, 0, True)
    torch.cuda.synchronize()
    nw = 8
    st = gd[:, L + 1 - nw:, :].reshape(B, nw, 8).flip(1)   # [B, wave, 8]
    m = st.mean(0)
    print('len', ln, 'phases', ln)
    for wv in range(nw):
        r = m[wv].tolist()
        if r[4] == 0: continue
        print(f' wave {wv}: fw body {r[0]/ln:7.0f} sync {r[1]/ln:7.0f} | bw body {r[2]/ln:7.0f} sync {r[3]/ln:7.0f}  cycles/phase | total {r[4]:9.0f} ticks')
    st2 = gd[:, L + 1 - 8 - nw:L + 1 - 8, :].reshape(B, nw, 8).flip(1).mean(0)
    names = ['stores', 'loads+terms+localmax', 'allreduce max', 'exp+sum', 'allreduce sum', 'log+fold', 'preamble+barrier(+bw!)', 'ptr setup+const loads']
    for wv in range(min(nw, 2)):
        print(f' wave {wv} fw segments (cycles/phase): ' + ', '.join(f'{names[k]}={st2[wv][k].item()/ln:.0f}' for k in range(0, 8)))
