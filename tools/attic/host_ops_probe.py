"""Host-side cost (enqueue time, no sync) of the small library products inside parser_ff's adjoint, per dtype."""
import time, torch
dev = torch.device('cuda:0')
def host_ms(fn, n=30):
    for _ in range(5): fn()
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(n): fn()
    t1 = time.perf_counter(); torch.cuda.synchronize(); return (t1 - t0) / n * 1e3, (time.perf_counter() - t0) / n * 1e3
for dt in (torch.bfloat16, torch.float32):
    H, nb, r, T = 256, 150, 16, 45
    W0s, W1s = torch.randn(4, nb, H, device=dev, dtype=dt), torch.randn(4, H, nb, device=dev, dtype=dt)
    b0s = torch.randn(4, nb, device=dev, dtype=dt); dbe = torch.randn(4, H, device=dev, dtype=dt); dWe = torch.randn(4, H, H, device=dev, dtype=dt)
    PW, dWp, W2 = torch.randn(6 * r, H, device=dev, dtype=dt), torch.randn(6 * r, H, device=dev, dtype=dt), torch.randn(H, H, device=dev, dtype=dt)
    gc, cm = torch.randn(256, H, device=dev, dtype=dt), torch.randn(256, 256, device=dev, dtype=dt)
    gs, inp = torch.randn(T, H, device=dev, dtype=dt), torch.randn(T, 32, device=dev, dtype=dt)
    ops = {
        "bmm W1s W0s": lambda: torch.bmm(W1s, W0s),
        "baddbmm dW1s": lambda: torch.baddbmm(torch.einsum("kh,kn->khn", dbe, b0s), dWe, W0s.transpose(1, 2)),
        "bmm dW0s": lambda: torch.bmm(W1s.transpose(1, 2), dWe),
        "bmm db0s": lambda: torch.bmm(W1s.transpose(1, 2), dbe.unsqueeze(2)),
        "PW.t() @ dWp": lambda: PW.t() @ dWp,
        "addmm dPW": lambda: torch.addmm(torch.outer(dWp[:, 0], W2[0]), dWp, W2.t()),
        "PW @ W2": lambda: PW @ W2,
        "gc.t() @ cmean": lambda: gc.t() @ cm,
        "gs.t() @ inp": lambda: gs.t() @ inp,
        "gs @ W(32)": lambda: gs @ torch.randn(H, 32, device=dev, dtype=dt),
    }
    for k, fn in ops.items():
        h, tot = host_ms(fn)
        print(f"{str(dt):15s} {k:18s} host {h:8.3f} ms   total {tot:8.3f} ms")
    Pb, b2, dbp = torch.randn(6 * 16, device=dev, dtype=dt), torch.randn(256, device=dev, dtype=dt), torch.randn(96, device=dev, dtype=dt)
    for k, fn in {"addmv": lambda: torch.addmv(Pb, PW, b2), "PW.t() @ dbp (mv)": lambda: PW.t() @ dbp, "outer": lambda: torch.outer(dbp, b2)}.items():
        h, tot = host_ms(fn)
        print(f"{str(dt):15s} {k:18s} host {h:8.3f} ms   total {tot:8.3f} ms")
