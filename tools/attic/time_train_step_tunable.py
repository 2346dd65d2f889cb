"""The training step with PyTorch's TunableOp choosing the library GEMM solutions (torch.cuda.tunable): eager warm-up tunes every GEMM shape
of the step once (results in gpurun_out/tunableop_results*.csv), then the step is captured and replayed as in tools/time_train_step.py."""
import sys, time, torch
sys.path.insert(0, '.'); sys.path.insert(0, 'tools')
import torch.cuda.tunable as tun
tun.enable(True)
tun.tuning_enable(True)
tun.set_max_tuning_duration(30)        # ms per solution
tun.set_max_tuning_iterations(20)
tun.set_filename("gpurun_out/tunableop_results.csv")
from vlgae_amd import train_step
dev = torch.device('cuda:0')
step = train_step.build(256, 40, 36, dev, dtype=torch.bfloat16)
t0 = time.perf_counter()
for _ in range(3): step()
torch.cuda.synchronize()
print('tuning + 3 steps: %.1f s' % (time.perf_counter() - t0))
tun.tuning_enable(False)
for _ in range(5): step()
torch.cuda.synchronize()
def wall(fn, n):
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(n): fn()
    torch.cuda.synchronize(); return (time.perf_counter() - t0) / n * 1e3
print('eager: %.3f ms/step' % wall(step, 30))
gr = torch.cuda.CUDAGraph()
s = torch.cuda.Stream()
s.wait_stream(torch.cuda.current_stream())
with torch.cuda.stream(s):
    for _ in range(3): step()
torch.cuda.current_stream().wait_stream(s)
with torch.cuda.graph(gr):
    out = step()
for _ in range(5): gr.replay()
print('graph: %.3f ms/step' % wall(gr.replay, 50))
print(len(tun.get_results()), 'tuned entries')
