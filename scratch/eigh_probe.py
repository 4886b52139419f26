import numpy as np, scipy.linalg as la, sys
sys.path.insert(0, "/root/repo")
from driftscan_amd._lib import Context
ctx = Context(0)
g = np.load("/root/repo/tests/golden/svdkl_unpol.npz")
S, N = g["m0_kl_cs"], g["m0_kl_cn"]
n = S.shape[0]
Lref = la.cholesky(N, lower=True)
dL = ctx.to_device(N[None].copy())
info = ctx.zpotrf(dL, n, n, stride=n*n, batch=1)
L = dL.cpu().numpy()[0]
print("info", info, "chol err", np.abs(L - Lref).max() / np.abs(Lref).max(), "recon", np.abs(L@L.conj().T - N).max()/np.abs(N).max())
Xref = la.solve_triangular(Lref, S, lower=True)
dX = ctx.to_device(S[None].copy())
ctx.ztrsm(dL, dX, n, n, n, n, conjtrans=False, batch=1)
ctx.sync()
X = dX.cpu().numpy()[0]
print("X err", np.abs(X - Xref).max() / np.abs(Xref).max(), "resid", np.abs(Lref@X - S).max()/np.abs(S).max())
Cref = la.solve_triangular(Lref, Xref.conj().T, lower=True)
dY = ctx.to_device(np.ascontiguousarray(Xref.conj().T)[None])
ctx.ztrsm(dL, dY, n, n, n, n, conjtrans=False, batch=1)
ctx.sync()
Y = dY.cpu().numpy()[0]
print("Y err", np.abs(Y - Cref).max() / np.abs(Cref).max())
print("eig of Cref", la.eigvalsh(0.5*(Cref+Cref.conj().T))[-3:], "eig of Y", la.eigvalsh(0.5*(Y+Y.conj().T))[-3:], "ref", g["m0_kl_evals"][-3:])
# row-wise errors of X
err = np.abs(X - Xref).max(axis=1) / np.abs(Xref).max(axis=1)
print("X row err (first 40)", np.round(np.log10(err[:40] + 1e-20), 1))
