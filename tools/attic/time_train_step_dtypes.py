import sys, time, torch
sys.path.insert(0, '.'); sys.path.insert(0, 'tools')
from vlgae_amd import train_step
torch.autograd.set_multithreading_enabled(False)
dev = torch.device('cuda:0')
for dt in (torch.bfloat16, torch.float32):
    step = train_step.build(256, 40, 36, dev, dtype=dt)
    for _ in range(5): step()
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(30): step()
    torch.cuda.synchronize(); print(dt, 'eager %.3f ms/step' % ((time.perf_counter() - t0) / 30 * 1e3))
