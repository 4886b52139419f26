import sys, time, numpy as np, torch
sys.path.insert(0, "/root/repo")
from driftscan_amd._lib import Context
ctx = Context(0, workspace_bytes=2 << 30)
M = N = K = 1024; batch = 16
A = torch.randn(batch, M, K, dtype=torch.complex128, device="cuda")
B = torch.randn(batch, K, N, dtype=torch.complex128, device="cuda")
C = torch.zeros(batch, M, N, dtype=torch.complex128, device="cuda")
for _ in range(3):
    ctx.zgemm(A, B, C, M, N, K, ldc=N, batch=batch, strideA=M*K, strideB=K*N, strideC=M*N, beta=0.0, alpha=1.0, rsA=K, csA=1, rsB=N, csB=1)
torch.cuda.synchronize()
