"""Weight-gradient products of the two encoders: vlg_linear_wgrad (split-K) vs the library, event-timed.
    python tools/time_wgrad_shapes.py"""
import sys, torch
sys.path.insert(0, '.')
from vlgae_amd import align
dev = torch.device('cuda:0')
def ev(fn, n=50):
    for _ in range(5): fn()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(n): fn()
    b.record(); torch.cuda.synchronize()
    return a.elapsed_time(b) / n * 1e3
for K, M, N, ld in ((9216, 256, 2048, 4096), (9216, 768, 2048, 4096), (2304, 768, 2048, 4096), (10240, 256, 800, 800), (256, 256, 2048, 4096), (64, 768, 2048, 4096)):
    dy = torch.randn(K, M, device=dev).bfloat16(); x = torch.randn(K, N, device=dev).bfloat16()
    out = torch.empty(M, ld, device=dev, dtype=torch.bfloat16)
    t_k = ev(lambda: align.linear_wgrad(dy, x, want_bias=False, out=(out[:, :N], None)))
    t_l = ev(lambda: torch.mm(dy.t(), x, out=out[:, :N]))
    outc = torch.empty(M, N, device=dev, dtype=torch.bfloat16)
    t_lc = ev(lambda: torch.mm(dy.t(), x, out=outc))
    ref = dy.float().t() @ x.float()
    align.linear_wgrad(dy, x, want_bias=False, out=(out[:, :N], None)); e1 = float((out[:, :N].float() - ref).abs().max() / ref.abs().max())
    torch.mm(dy.t(), x, out=out[:, :N]); e2 = float((out[:, :N].float() - ref).abs().max() / ref.abs().max())
    print(f"K={K} M={M} N={N}: split-K {t_k:.1f} us (err {e1:.1e}), library strided-out {t_l:.1f} us (err {e2:.1e}), library contiguous {t_lc:.1f} us; {2*K*M*N/1e9:.1f} GFLOP")
for K, M, N in ((40960, 256, 256), (40960, 512, 256), (10496, 384, 256), (41000, 32, 256)):
    dy = torch.randn(K, M, device=dev).bfloat16(); x = torch.randn(K, N, device=dev).bfloat16()
    t_k = ev(lambda: align.linear_wgrad(dy, x, out_dtype=torch.bfloat16))
    t_l = ev(lambda: torch.mm(dy.t(), x))
    print(f"K={K} M={M} N={N}: split-K (+bias) {t_k:.1f} us, library {t_l:.1f} us; {2*K*M*N/1e9:.1f} GFLOP")

# float32 operands (three bf16 products per pair) against the library's fp32 GEMM
for K, M, N in ((9216, 256, 2048), (10240, 256, 800), (40960, 256, 256), (40960, 512, 256), (10496, 384, 256), (10496, 128, 128), (41000, 32, 256), (256, 256, 2048)):
    dy = torch.randn(K, M, device=dev) * 1e-3; x = torch.randn(K, N, device=dev)
    t_k = ev(lambda: align.linear_wgrad(dy, x))
    t_l = ev(lambda: (torch.mm(dy.t(), x), dy.sum(0)))
    ref = dy.double().t() @ x.double()
    e1 = float((align.linear_wgrad(dy, x)[0].double() - ref).abs().max() / ref.abs().max()); e2 = float((torch.mm(dy.t(), x).double() - ref).abs().max() / ref.abs().max())
    print(f"f32 K={K} M={M} N={N}: split-K x3 (+bias) {t_k:.1f} us (err {e1:.1e}), library fp32 {t_l:.1f} us (err {e2:.1e})")
