"""Eager vs HIP-graph timing of the training step (vlgae_amd/train_step.py) + host-side profile.
    python tools/time_train_step.py [B L R] [--f32] [--r3] [--shipped: factors rel attr img, i.e. 1369 columns at R = 36] [--profile]"""
import sys, time, torch
sys.path.insert(0, '.'); sys.path.insert(0, 'tools')
from vlgae_amd import train_step
dev = torch.device('cuda:0')
pos = [a for a in sys.argv[1:] if not a.startswith('--')]
B, L, V = (int(pos[0]), int(pos[1]), int(pos[2])) if len(pos) > 2 else (256, 40, 36)
kw = dict(factors=('rel', 'attr', 'img')) if '--shipped' in sys.argv else {}
step = train_step.build(B, L, V, dev, wiring='r3' if '--r3' in sys.argv else 'reference', dtype=torch.float32 if '--f32' in sys.argv else torch.bfloat16, **kw)
for _ in range(5): res = step()
torch.cuda.synchronize()
import hashlib
print('sha256(loss|grads) %s' % hashlib.sha256(b''.join(t.detach().float().cpu().numpy().tobytes() for t in [res[0]] + [res[1][k] for k in sorted(res[1])])).hexdigest()[:16])
def wall(fn, n):
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(n): fn()
    torch.cuda.synchronize(); return (time.perf_counter() - t0) / n * 1e3
print('eager: %.3f ms/step' % wall(step, 30))
t0 = time.perf_counter()
for _ in range(30): step()
print('host enqueue: %.3f ms/step' % ((time.perf_counter() - t0) / 30 * 1e3)); torch.cuda.synchronize()
try:
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
except Exception as e:
    print('graph capture failed:', repr(e)[:300])
if '--profile' in sys.argv:
    import cProfile, pstats
    pr = cProfile.Profile(); pr.enable()
    for _ in range(20): step()
    pr.disable(); torch.cuda.synchronize()
    pstats.Stats(pr).sort_stats('cumulative').print_stats(35)
