"""Reproducer hunt: batches of Gram matrices with exactly zero rows / columns through the one-stage and the two-stage
tridiagonalisation (DM_TRD_TWOSTAGE=0/1 per process), eigenvalues against numpy."""
import os, sys, numpy as np, torch
sys.path.insert(0, "/root/repo")
from driftscan_amd import device
ctx = device.get_context(workspace_bytes=16 << 30)
rng = np.random.default_rng(5)
def batch(nb, n, K, frac_zero):
    out = np.zeros((nb, n, n), dtype=np.complex128)
    for b in range(nb):
        A = rng.standard_normal((n, K)) + 1j * rng.standard_normal((n, K))
        A *= np.logspace(0, -8, n)[:, None]                      # graded rows
        nz = int(frac_zero[b % len(frac_zero)] * n)
        if nz:
            idx = rng.permutation(n)[:nz] if (b % 2) else np.arange(n - nz, n)
            A[idx] = 0.0
        out[b] = A @ A.conj().T
    return out
for nb, n in ((48, 452), (200, 452), (374, 452), (374, 448)):
    G = batch(nb, n, 864, (0.0, 0.5, 0.8, 0.95))
    ref = np.linalg.eigvalsh(G)
    ev, W = ctx.herm_eig(ctx.to_device(G.copy()), n, n, strideC=n * n, batch=nb)
    got = np.sort(ev.cpu().numpy()[:, :n], axis=1)
    err = np.abs(got - ref).max(axis=1) / np.abs(ref).max(axis=1)
    Wh = W.cpu().numpy()
    un = max(np.abs(Wh[b] @ Wh[b].conj().T - np.eye(n)).max() for b in range(0, nb, max(1, nb // 8)))
    print("TWOSTAGE=%s nb %d n %d: max eig err %.2e (bad matrices %d), unitarity %.2e" % (os.environ.get("DM_TRD_TWOSTAGE"), nb, n, err.max(), (err > 1e-10).sum(), un), flush=True)
