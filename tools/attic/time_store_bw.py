"""What this GPU sustains for a pure 774 MB fp32 store stream (the size of the materialised [B,A,Q,V] alignment tensor at config-2): torch's
fill_ / zero_ (hipMemsetAsync) / a copy, event-timed -- the ceiling `align_full_kernel` (0.164-0.185 ms = 4.2-4.7 TB/s) is measured against.
    python tools/time_store_bw.py"""
import torch
dev = torch.device('cuda:0')
n = 256 * 256 * 82 * 36
x = torch.empty(n, device=dev)
y = torch.randn(n, device=dev)
def ev(fn, k=30):
    for _ in range(5): fn()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(k): fn()
    b.record(); torch.cuda.synchronize()
    return a.elapsed_time(b) / k
for name, fn, nbytes in (("fill_(1.5)", lambda: x.fill_(1.5), 4 * n), ("zero_()", lambda: x.zero_(), 4 * n), ("copy_ (read + write)", lambda: x.copy_(y), 8 * n),
                         ("mul_ (read + write in place)", lambda: x.mul_(1.0001), 8 * n)):
    ms = ev(fn)
    print(f"{name}: {ms * 1e3:.1f} us, {nbytes / ms / 1e9:.2f} TB/s moved ({4 * n / ms / 1e9:.2f} TB/s of stores)")

# pure READ streams of the same 774 MB (the cotangent of the materialised tensor is read once per gradient by the a9 backward kernels)
for name, fn in (("sum() (read)", lambda: y.sum()), ("max() (read)", lambda: y.max()), ("dot(y, y) (read twice the bytes? no: one tensor, two operands)", lambda: torch.dot(y, y))):
    ms = ev(fn)
    print(f"{name}: {ms * 1e3:.1f} us, {4 * n / ms / 1e9:.2f} TB/s of reads")
