"""Batched Hermitian eigensolver timing: python scratch/eig_bench.py n batch [n batch ...]"""
import sys, time
import numpy as np, torch
sys.path.insert(0, "/root/repo")
from driftscan_amd._lib import Context
ctx = Context(0, workspace_bytes=48 << 30)
args = [int(x) for x in sys.argv[1:]] or [864, 512, 4096, 4]
for n, nb in zip(args[::2], args[1::2]):
    g = torch.Generator(device="cuda").manual_seed(1)
    X = torch.randn(nb, n, n, dtype=torch.complex128, device="cuda", generator=g)
    C0 = X @ X.conj().transpose(1, 2) / n
    del X
    ts = []
    for rep in range(3):
        C = C0.clone()
        torch.cuda.synchronize(); t0 = time.perf_counter()
        ev, W = ctx.herm_eig(C, n, n, strideC=n * n, batch=nb)
        torch.cuda.synchronize(); ts.append(time.perf_counter() - t0)
        del C, W
    print("n %5d batch %4d: %s s" % (n, nb, " ".join("%.3f" % t for t in ts)), flush=True)
    del C0
    torch.cuda.empty_cache()
