"""One-stage vs two-stage tridiagonalisation over (n, batch): seconds per call of the Hermitian eigensolver (all vectors)."""
import os, sys, time
sys.path.insert(0, os.path.join(os.path.dirname(__file__), ".."))
import numpy as np
import torch
from driftscan_amd._lib import Context
cases = [(200, 128), (300, 64), (500, 128), (864, 512), (1200, 32), (2000, 8), (3000, 8), (4000, 8), (6000, 8), (8192, 1), (16384, 1)]
if len(sys.argv) > 1:
    cases = [tuple(int(x) for x in a.split("x")) for a in sys.argv[1:]]
ctx = Context(0, workspace_bytes=int(float(os.environ.get("WS_GB", "120")) * (1 << 30)))
for n, nb in cases:
    g = torch.Generator(device="cuda").manual_seed(3)
    A = torch.randn((nb, n, n), dtype=torch.float64, device="cuda", generator=g) + 1j * torch.randn((nb, n, n), dtype=torch.float64, device="cuda", generator=g)
    A = torch.triu(A + A.conj().transpose(1, 2)).contiguous()
    out = []
    for mode in ("0", "1"):
        os.environ["DM_TRD_TWOSTAGE"] = mode
        best = 1e9
        for rep in range(3):
            C = A.clone()
            torch.cuda.synchronize()
            t0 = time.time()
            ev, W = ctx.herm_eig(C, n, n, strideC=n * n, batch=nb)
            ctx.sync(); torch.cuda.synchronize()
            best = min(best, time.time() - t0)
        out.append(best)
        del ev, W, C
    print("n %6d batch %4d  one-stage %8.4f s  two-stage %8.4f s  ratio %.2f" % (n, nb, out[0], out[1], out[0] / out[1]), flush=True)
    del A
