"""Bulge-chase timing probe: one matrix (the dependent chain alone) and a batch, through herm_eig with the two-stage path forced.
usage: chase_probe.py [n:batch ...]"""
import os
import sys

os.environ["DM_TRD_TWOSTAGE"] = "1"
sys.path.insert(0, os.path.join(os.path.dirname(__file__), ".."))
import numpy as np

from driftscan_amd._lib import Context

ctx = Context(0, workspace_bytes=int(os.environ.get("PROBE_WS_GB", "16")) << 30)
rng = np.random.default_rng(5)
cases = [tuple(int(x) for x in a.split(":")) for a in sys.argv[1:]] or [(1218, 1), (1218, 8), (700, 128), (4000, 1), (8000, 1)]
for n, nb in cases:
    A = rng.standard_normal((nb, n, n)) + 1j * rng.standard_normal((nb, n, n))
    A = A + A.conj().transpose(0, 2, 1)
    dA = ctx.to_device(np.triu(A))
    for rep in range(3):
        ctx.prof_reset(1)
        ev, W = ctx.herm_eig(dA.clone(), n, n, strideC=n * n, batch=nb)
        ctx.sync()
        pr = ctx.prof_report()
    ch = pr.get("sb_chase", {})
    ref = np.linalg.eigvalsh(A[0])
    err = np.abs(np.sort(ev.cpu().numpy().reshape(nb, n)[0]) - ref).max() / np.abs(ref).max()
    print("n %5d batch %3d  sb_chase %.3f ms (%.2f us per sweep of the largest matrix)  q2 %.3f ms  evals err %.1e"
          % (n, nb, ch.get("ms", 0.0), 1e3 * ch.get("ms", 0.0) / max(n - 1, 1), pr.get("sb_q2_apply", {}).get("ms", 0.0), err), flush=True)
