import sys, time, numpy as np, torch
sys.path.insert(0, '/root/repo'); sys.path.insert(0, '/root/repo/tests/golden')
import _ref_import, oracle
ts = _ref_import.import_torch_struct()
torch.set_num_threads(8)
def synth(B, L):
    g = torch.Generator().manual_seed(0)
    dec = torch.randn(B, L, 2, 2, 2, generator=g).log_softmax(-1); attach = torch.randn(B, L, L, 2, generator=g); root = torch.randn(B, L, generator=g).log_softmax(-1)
    return dec, attach, root
for B, L, reps in ((256, 40, 3), (256, 80, 2)):
    dec, attach, root = synth(B, L)
    lengths = torch.full((B,), L, dtype=torch.long)
    best = 1e9
    for _ in range(reps):
        md, ma = ts.DMV1o.merge(dec, attach, root)
        d, a = md.requires_grad_(), ma.requires_grad_()
        t0 = time.perf_counter()
        z = ts.DMV1o([d, a], lengths).partition.sum()
        torch.autograd.grad(z, [d, a])
        best = min(best, time.perf_counter() - t0)
    omd, oma = oracle.dmv1o_merge(dec.numpy(), attach.numpy(), root.numpy())
    ob = 1e9
    for _ in range(5):
        t0 = time.perf_counter(); oracle.dmv1o(omd, oma, lengths.numpy(), 'log', np.float32); ob = min(ob, time.perf_counter() - t0)
    oracle.set_threads(1)
    t0 = time.perf_counter(); oracle.dmv1o(omd, oma, lengths.numpy(), 'log', np.float32); o1 = time.perf_counter() - t0
    oracle.set_threads(8)
    print(f'DMV1o fwd+bwd B={B} L={L}: reference {best*1e3:.0f} ms | C oracle 8 thr {ob*1e3:.0f} ms | C oracle 1 thr {o1*1e3:.0f} ms')
# alignment
src, joint = _ref_import.import_joint()
B = A = 256; Q, V, d = 82, 36, 128
g = torch.Generator().manual_seed(0)
txt = torch.randn(B, Q, d, generator=g); vis = torch.randn(A, V, d, generator=g)
tm = torch.ones(B, Q, dtype=torch.bool); vm = torch.ones(A, V, dtype=torch.bool)
import warnings; warnings.filterwarnings('ignore')
best = 1e9
for _ in range(3):
    t0 = time.perf_counter()
    joint.DependencyBoxRel.gather_logit_simple(None, None, (vis.refine_names('A','V','Y'), vm.refine_names('A','V'), None), (txt.refine_names('B','Q','X'), tm.refine_names('B','Q'), None), None)
    best = min(best, time.perf_counter() - t0)
t0 = time.perf_counter(); oracle.bilinear_align(txt.numpy(), vis.numpy(), tm.numpy(), vm.numpy()); ob = time.perf_counter() - t0
print(f'gather_logit fwd B=A=256 Q=82 V=36 d=128: reference {best*1e3:.0f} ms | C oracle 8 thr {ob*1e3:.0f} ms')
