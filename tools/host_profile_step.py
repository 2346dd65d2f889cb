import sys, time, torch
sys.path.insert(0, '.'); sys.path.insert(0, 'tools')
from vlgae_amd import train_step
dev = torch.device('cuda:0')
step = train_step.build(256, 40, 36, dev, wiring='reference', dtype=torch.bfloat16)
for _ in range(5): step()
torch.cuda.synchronize()
import cProfile, pstats
pr = cProfile.Profile(); pr.enable()
for _ in range(20): step()
pr.disable(); torch.cuda.synchronize()
st = pstats.Stats(pr)
st.sort_stats('tottime').print_stats(45)
st.sort_stats('cumulative').print_stats(60)
