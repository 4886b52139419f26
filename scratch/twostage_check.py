"""Two-stage tridiagonalisation: stage-by-stage check against numpy (development aid).

DM_TRD_TWOSTAGE=1 is set here; DM_SB_DUMP makes the library write the band after the first stage and the
tridiagonal after the bulge chase, whose spectra must equal the spectrum of the input."""
import os
import sys

os.environ["DM_TRD_TWOSTAGE"] = "1"
os.environ["DM_SB_DUMP"] = "/tmp/sbdump"
sys.path.insert(0, os.path.join(os.path.dirname(__file__), ".."))
import numpy as np
import scipy.linalg as sla

from driftscan_amd._lib import Context

SB = 32
ctx = Context(0, workspace_bytes=4 << 30)
rng = np.random.default_rng(5)
sizes = [int(a) for a in sys.argv[1:]] or [97, 100, 130, 161, 200, 515]
ok = True
for n in sizes:
    nb = 3
    A = rng.standard_normal((nb, n, n)) + 1j * rng.standard_normal((nb, n, n))
    A = A + A.conj().transpose(0, 2, 1)
    A[1] *= np.logspace(0, -8, n)[:, None] * np.logspace(0, -8, n)[None, :]  # graded
    ref = np.linalg.eigvalsh(A)
    ev, W = ctx.herm_eig(ctx.to_device(np.triu(A)), n, n, strideC=n * n, batch=nb)
    ctx.sync()
    ev = ev.cpu().numpy().reshape(nb, n)
    W = W.cpu().numpy().reshape(nb, n, n)
    band = np.fromfile("/tmp/sbdump.band", dtype=np.complex128).reshape(nb, n, 2 * SB)
    d = np.fromfile("/tmp/sbdump.d").reshape(nb, n)
    e = np.fromfile("/tmp/sbdump.e").reshape(nb, n)
    for b in range(nb):
        sc = np.abs(ref[b]).max()
        Bm = np.zeros((n, n), complex)
        for i in range(SB + 1):
            idx = np.arange(n - i)
            Bm[idx + i, idx] = band[b, : n - i, i]
        Bm = np.tril(Bm) + np.tril(Bm, -1).conj().T
        eb = np.linalg.eigvalsh(Bm)
        et = sla.eigvalsh_tridiagonal(d[b], e[b, : n - 1]) if n > 1 else d[b]
        es = np.sort(ev[b])
        V = W[b].conj().T
        order = np.argsort(ev[b])
        res = np.abs(A[b] @ V - V * ev[b][None, :]).max() / sc
        orth = np.abs(V.conj().T @ V - np.eye(n)).max()
        line = (n, b, np.abs(eb - ref[b]).max() / sc, np.abs(et - ref[b]).max() / sc, np.abs(es - ref[b]).max() / sc, res, orth)
        print("n %4d b %d  band %.1e  tridiag %.1e  evals %.1e  resid %.1e  orth %.1e" % line)
        if max(line[2:]) > 1e-11:
            ok = False
print("OK" if ok else "FAILED")
sys.exit(0 if ok else 1)
