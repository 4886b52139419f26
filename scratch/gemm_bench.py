import sys, time, numpy as np, torch
sys.path.insert(0, "/root/repo")
from driftscan_amd._lib import Context
ctx = Context(0, workspace_bytes=2 << 30)
def run(M, N, K, batch, mode="NN", reps=5):
    A = torch.randn(batch, M, K, dtype=torch.complex128, device="cuda")
    B = torch.randn(batch, K, N, dtype=torch.complex128, device="cuda")
    C = torch.zeros(batch, M, N, dtype=torch.complex128, device="cuda")
    kw = dict(rsA=K, csA=1, rsB=N, csB=1) if mode == "NN" else dict(rsA=K, csA=1, rsB=1, csB=K, conjB=True)
    if mode == "NC":
        B = torch.randn(batch, N, K, dtype=torch.complex128, device="cuda")
    for _ in range(2):
        ctx.zgemm(A, B, C, M, N, K, ldc=N, batch=batch, strideA=M*K, strideB=K*N, strideC=M*N, beta=1.0, alpha=-1.0, **kw)
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(reps):
        ctx.zgemm(A, B, C, M, N, K, ldc=N, batch=batch, strideA=M*K, strideB=K*N, strideC=M*N, beta=1.0, alpha=-1.0, **kw)
    torch.cuda.synchronize(); dt = (time.perf_counter() - t0) / reps
    fl = 8.0 * M * N * K * batch
    print(f"{mode} M={M:5d} N={N:5d} K={K:5d} batch={batch:4d}: {dt*1e3:8.3f} ms  {fl/dt/1e12:6.2f} TFLOP/s")
run(2048, 2048, 2048, 1)
run(1024, 1024, 1024, 16)
run(1024, 1024, 1024, 16, "NC")
run(64, 1024, 1024, 128)
run(1024, 1024, 64, 128)
run(1024, 1024, 64, 128, "NC")
run(92, 92, 129, 4096, "NC")
run(32, 1024, 512, 128)
run(4096, 4096, 128, 4)
run(128, 4096, 4096, 4)
run(4096, 1024, 128, 8)
run(92, 4989, 413, 64, "NC")
run(128, 4989, 413, 64, "NC")
