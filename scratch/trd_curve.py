"""Per-column efficiency of trd_symv / trd_wx: a batch of NB random Hermitian n x n matrices through the tridiagonal
eigensolver under `rocprofv3 --kernel-trace`; scratch/trd_curve_post.py turns the trace into GB/s against n - k."""
import sys

import os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np

from driftscan_amd._lib import Context

n = int(sys.argv[1]) if len(sys.argv) > 1 else 1024
nb = int(sys.argv[2]) if len(sys.argv) > 2 else 129
ctx = Context(0, workspace_bytes=24 << 30)
rng = np.random.default_rng(0)
C = rng.standard_normal((nb, n, n)) + 1j * rng.standard_normal((nb, n, n))
C = C + C.conj().transpose(0, 2, 1)
dC = ctx.to_device(C)
for rep in range(2):
    ev, W = ctx.herm_eig(dC.clone(), n, n, strideC=n * n, batch=nb)
    ctx.sync()
print("done", float(ev.cpu().numpy().max()))
