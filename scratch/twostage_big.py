"""One large Hermitian eigenproblem: one-stage vs two-stage tridiagonalisation (time, residual, orthogonality)."""
import os, sys, time
sys.path.insert(0, os.path.join(os.path.dirname(__file__), ".."))
import numpy as np
import torch
from driftscan_amd._lib import Context
n = int(sys.argv[1]) if len(sys.argv) > 1 else 8192
nb = int(sys.argv[2]) if len(sys.argv) > 2 else 1
modes = sys.argv[3].split(",") if len(sys.argv) > 3 else ["1", "0"]
ctx = Context(0, workspace_bytes=int(float(os.environ.get("WS_GB", "60")) * (1 << 30)))
g = torch.Generator(device="cuda").manual_seed(3)
A = torch.randn((nb, n, n), dtype=torch.float64, device="cuda", generator=g) + 1j * torch.randn((nb, n, n), dtype=torch.float64, device="cuda", generator=g)
A = A + A.conj().transpose(1, 2)
for mode in modes:
    os.environ["DM_TRD_TWOSTAGE"] = mode
    for rep in range(2):
        C = torch.triu(A).contiguous()
        torch.cuda.synchronize()
        t0 = time.time()
        ev, W = ctx.herm_eig(C, n, n, strideC=n * n, batch=nb)
        ctx.sync(); torch.cuda.synchronize()
        dt = time.time() - t0
    V = W[0].conj().T
    res = (A[0] @ V - V * ev[0][None, :]).abs().max().item() / ev[0].abs().max().item()
    orth = (V.conj().T @ V - torch.eye(n, dtype=V.dtype, device="cuda")).abs().max().item()
    print("n %d nb %d twostage %s  %.3f s  resid %.2e orth %.2e" % (n, nb, mode, dt, res, orth), flush=True)
    del V, W, ev, C
