import sys, numpy as np, torch
sys.path.insert(0, "/root/repo")
from driftscan_amd import device
from driftscan_amd import kltransform
ctx = device.get_context(workspace_bytes=4 << 30)
rng = np.random.default_rng(1)
for n in (20, 42, 66, 80, 100, 159, 300):
    X = rng.standard_normal((n, n)) + 1j * rng.standard_normal((n, n))
    A0 = X @ X.conj().T
    for scale in (1.0, 1e-9, 1e-18):
        A = A0 * scale
        ref = np.linalg.eigvalsh(A)
        ev, W = ctx.herm_eig(ctx.to_device(A[None].copy()), n, n)
        got = np.sort(ev.cpu().numpy()[0][:n])
        e1 = np.abs(got - ref).max() / np.abs(ref).max()
        # generalised problem with B = I
        ev2, _, _ = kltransform.eigh_gen(A, np.eye(n, dtype=np.complex128))
        e2 = np.abs(np.sort(ev2) - ref).max() / np.abs(ref).max()
        # graded spectrum: eigenvalues spread over 12 decades
        print("n %4d scale %.0e: herm_eig rel err %.2e   eigh_gen(A, I) rel err %.2e" % (n, scale, e1, e2), flush=True)
