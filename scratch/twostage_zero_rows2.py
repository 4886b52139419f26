import os, sys, numpy as np, torch
sys.path.insert(0, "/root/repo")
from driftscan_amd import device
from driftscan_amd._lib import DriftMIError
ctx = device.get_context(workspace_bytes=8 << 30)
rng = np.random.default_rng(5)
def mk(n, K, nz, pattern, graded):
    A = rng.standard_normal((n, K)) + 1j * rng.standard_normal((n, K))
    if graded: A *= np.logspace(0, -8, n)[:, None]
    if nz:
        idx = {"tail": np.arange(n - nz, n), "head": np.arange(nz), "random": rng.permutation(n)[:nz], "middle": np.arange(max(0, (n - nz) // 2), max(0, (n - nz) // 2) + nz)}[pattern]
        A[idx] = 0.0
    return A @ A.conj().T
for n in (200, 452):
    for graded in (False, True):
        for pattern in ("tail", "head", "random", "middle"):
            for fz in (0.1, 0.5, 0.95):
                nz = int(fz * n)
                G = np.stack([mk(n, 864, nz, pattern, graded) for _ in range(4)])
                ref = np.linalg.eigvalsh(G)
                try:
                    ev, W = ctx.herm_eig(ctx.to_device(G.copy()), n, n, strideC=n * n, batch=4)
                    got = np.sort(ev.cpu().numpy()[:, :n], axis=1)
                    err = np.abs(got - ref).max() / np.abs(ref).max()
                    res = "err %.1e" % err
                except DriftMIError as e:
                    res = "FAIL " + str(e)[-40:]
                print("n %d graded %d %-7s zero %.2f: %s" % (n, graded, pattern, fz, res), flush=True)
