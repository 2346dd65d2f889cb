"""One DP workload of the bench line as a stand-alone program (what rocprofv3 is pointed at for the secondary DP lines):
    python tools/dp_workload.py B L {bf16|f32} [launches]
runs the fused DMV1o inside+outside kernel (Log) on the bench's synthetic potentials through the raw C ABI."""
import sys
import torch
sys.path.insert(0, '.')
from vlgae_amd.bench import secondary
B, L, dt = int(sys.argv[1]), int(sys.argv[2]), sys.argv[3]
n = int(sys.argv[4]) if len(sys.argv) > 4 else 20
dev = torch.device('cuda:0')
launch = secondary.dp_raw(B, L, torch.bfloat16 if dt == 'bf16' else torch.float32, dev)
for _ in range(n):
    launch()
torch.cuda.synchronize()
print(f"dmv1o_B{B}_L{L}_{dt}: {n} launches, workspace {launch.ws_bytes} bytes")
