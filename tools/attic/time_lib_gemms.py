"""The parser feed-forwards' library GEMMs (bf16, bias in the epilogue) under torch's two BLAS back ends, against their HBM floor."""
import sys, torch
dev = torch.device('cuda:0')
def ev(fn, n=50):
    for _ in range(10): fn()
    torch.cuda.synchronize(); e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n): fn()
    e1.record(); torch.cuda.synchronize(); return e0.elapsed_time(e1) * 1e3 / n
g = torch.Generator().manual_seed(0)
shapes = [("A1 = X Wnh^T", 10288, 256, 512), ("A2 = A1 Wv^T", 20576, 256, 256), ("Z = A2 Wlr^T", 20576, 256, 512), ("A4 = A3 Wd^T", 41152, 256, 256),
          ("big = A5 Wp^T", 40960, 256, 32), ("X = emb We^T", 10240, 800, 256), ("g W (dgrad)", 41152, 256, 256)]
for lib in ("default", "cublaslt", "cublas"):
    if lib != "default":
        try:
            torch.backends.cuda.preferred_blas_library(lib)
        except Exception as e:
            print(lib, "unavailable:", e); continue
    print("back end:", lib, torch.backends.cuda.preferred_blas_library())
    for name, M, K, N in shapes:
        a = torch.randn(M, K, generator=g).to(dev, torch.bfloat16); w = torch.randn(N, K, generator=g).to(dev, torch.bfloat16); b = torch.randn(N, generator=g).to(dev, torch.bfloat16)
        t1 = ev(lambda: torch.addmm(b, a, w.t()))
        wt = w.t().contiguous()
        t2 = ev(lambda: a @ w)  if K == N else float('nan')           # dgrad form: [M,K] @ [K,N] with w row-major [N(=K), K]
        floor = (M * K + M * N) * 2 / 5e6   # us at 5 TB/s
        print(f"  {name:16s} [{M},{K}]x[{K},{N}]: addmm {t1:6.1f} us  (dgrad form {t2:6.1f})  HBM floor {floor:5.1f} us", flush=True)
