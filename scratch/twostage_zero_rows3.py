"""The failing case of twostage_zero_rows2.py alone (n = 452, random zero rows, 95 %), with the band/tridiagonal dump of
DM_SB_DUMP looked at afterwards: where do the non-finite / non-converging values come from?"""
import os, sys, numpy as np
os.environ["DM_TRD_TWOSTAGE"] = "1"
os.environ["DM_SB_DUMP"] = "/root/repo/gpurun_out/sbdump"
sys.path.insert(0, "/root/repo")
from driftscan_amd import device
from driftscan_amd._lib import DriftMIError
ctx = device.get_context(workspace_bytes=8 << 30)
rng = np.random.default_rng(5)
n, K = 452, 864
nb = int(os.environ.get("NB", "4"))
Gs = []
for _ in range(nb):
    A = rng.standard_normal((n, K)) + 1j * rng.standard_normal((n, K))
    A[rng.permutation(n)[: int(0.95 * n)]] = 0.0
    Gs.append(A @ A.conj().T)
G = np.stack(Gs)
try:
    ev, W = ctx.herm_eig(ctx.to_device(G.copy()), n, n, strideC=n * n, batch=nb)
    got = np.sort(ev.cpu().numpy()[:, :n], axis=1)
    print("ok, err", np.abs(got - np.linalg.eigvalsh(G)).max() / np.abs(G).max())
except DriftMIError as e:
    print("FAIL", e)
d = np.fromfile("/root/repo/gpurun_out/sbdump.d")
e = np.fromfile("/root/repo/gpurun_out/sbdump.e")
band = np.fromfile("/root/repo/gpurun_out/sbdump.band", dtype=np.complex128)
print("d", d.shape, "finite", np.isfinite(d).all(), "e finite", np.isfinite(e).all(), "band finite", np.isfinite(band).all(), band.shape)
for p in range(nb):
    dp, ep = d[p * n:(p + 1) * n], e[p * n:(p + 1) * n]
    T = np.diag(dp) + np.diag(ep[: n - 1], 1) + np.diag(ep[: n - 1], -1)
    if np.isfinite(T).all():
        w = np.linalg.eigvalsh(T)
        print(p, "tridiagonal eigenvalue error", np.abs(w - np.linalg.eigvalsh(G[p])).max() / np.abs(G[p]).max(), "nonzero e", np.count_nonzero(ep), "max|e|", np.abs(ep).max(), "min nonzero |e|", np.abs(ep[ep != 0]).min() if np.count_nonzero(ep) else 0)
    else:
        print(p, "non-finite at d", np.where(~np.isfinite(dp))[0][:10], "e", np.where(~np.isfinite(ep))[0][:10])
