"""What this box's HBM delivers to simple torch kernels (calibration for the streaming kernels' GB/s figures)."""
import torch, time
dev = torch.device('cuda:0')
x = torch.randn(256, 256, 82, 36, device=dev)   # 774 MB fp32
y = torch.empty_like(x)
def t(fn, n=10):
    for _ in range(3): fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n
nb = x.numel() * 4
for name, fn, traffic in (("sum (read)", lambda: x.sum(), nb), ("copy (read + write)", lambda: y.copy_(x), 2 * nb),
                          ("fill (write)", lambda: y.fill_(1.0), nb), ("sum over last two dims (read)", lambda: x.sum((2, 3)), nb)):
    ms = t(fn)
    print("%-32s %.3f ms  %.2f TB/s" % (name, ms, traffic / ms / 1e9))
