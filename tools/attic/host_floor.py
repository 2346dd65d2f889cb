"""Floor of the torch autograd path for one custom Function with two differentiable inputs on this host: what
`autograd.grad(F.apply(a, b).sum(), [a, b])` costs when forward / backward do no work at all."""
import time, torch
dev = torch.device("cuda:0")
B = 256
a = torch.zeros(B, 41, 2, 2, 2, device=dev, dtype=torch.bfloat16, requires_grad=True)
b = torch.zeros(B, 41, 41, 2, device=dev, dtype=torch.bfloat16, requires_grad=True)
out = torch.zeros(B, 1, device=dev)
ga, gb = torch.zeros_like(a), torch.zeros_like(b)
class Nop(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x, y):
        return out.clone()
    @staticmethod
    def backward(ctx, g):
        return ga, gb
def t(fn, n=300):
    for _ in range(30): fn()
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(n): fn()
    t1 = time.perf_counter(); torch.cuda.synchronize()
    return (t1 - t0) / n * 1e6
print("apply only            %.1f us" % t(lambda: Nop.apply(a, b)))
print("apply + sum           %.1f us" % t(lambda: Nop.apply(a, b).sum()))
print("apply + sum + grad    %.1f us" % t(lambda: torch.autograd.grad(Nop.apply(a, b).sum(), [a, b])))
def bw():
    a.grad = b.grad = None
    Nop.apply(a, b).sum().backward()
print("apply + sum.backward  %.1f us" % t(bw))
z = Nop.apply(a, b)
print("grad w/ ones, no sum  %.1f us" % t(lambda: torch.autograd.grad(Nop.apply(a, b), [a, b], out)))
