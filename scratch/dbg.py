import numpy as np, scipy.linalg as la, sys
sys.path.insert(0,'/root/repo/scratch')
from proto_jacobi import block_jacobi_rows
g = np.load("/root/repo/tests/golden/svdkl_unpol.npz")
cs, cn = g["m0_kl_cs"], g["m0_kl_cn"]
n = cs.shape[0]
L = la.cholesky(cn, lower=True)
X = la.solve_triangular(L, cs, lower=True)
C = la.solve_triangular(L, X.conj().T, lower=True).conj().T
print("herm err", np.abs(C-C.conj().T).max(), np.abs(C).max())
C = 0.5*(C+C.conj().T)
w = la.eigvalsh(C)
print("eigvalsh(C) top", w[-4:], "ref", g["m0_kl_evals"][-4:], "min", w[:3])
Z = np.concatenate([C, np.eye(n)], axis=1)
Zr, s, sw = block_jacobi_rows(Z, np.arange(n), 8, verbose=True, maxsweeps=12)
print("s top", s[:4])
# diagnose: after 12 sweeps, gram of rows
Y = Zr[:, :n]
G = Y@Y.conj().T
d = np.sqrt(np.diag(G).real)
off = np.abs(G)/np.outer(d,d); np.fill_diagonal(off,0)
i,j = np.unravel_index(off.argmax(), off.shape)
print("worst pair", i, j, off[i,j], d[i], d[j])
print("norms", d[:10], d[-10:])
# try scalar tolerance: treat rows with norm < eps*max as converged
