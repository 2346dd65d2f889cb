"""One shape per process for rocprofv3: python tools/time_gemm2.py K M N"""
import sys, torch
sys.path.insert(0, '.')
from vlgae_amd import align
K, M, N = map(int, sys.argv[1:4])
dev = torch.device('cuda:0'); g = torch.Generator().manual_seed(0)
dy = torch.randn(K, M, generator=g).to(dev).bfloat16(); x = torch.randn(K, N, generator=g).to(dev).bfloat16()
for _ in range(30): align.linear_wgrad(dy, x)
torch.cuda.synchronize()
