import numpy as np, scipy.linalg as la, sys
sys.path.insert(0,'/root/repo/scratch')
import proto_jacobi2 as pj
g = np.load("/root/repo/tests/golden/svdkl_unpol.npz")
cs, cn = g["m0_kl_cs"], g["m0_kl_cn"]
n = cs.shape[0]
L = la.cholesky(cn, lower=True)
X = la.solve_triangular(L, cs, lower=True)
C = la.solve_triangular(L, X.conj().T, lower=True).conj().T
C = 0.5*(C+C.conj().T)
Z = np.concatenate([C, np.eye(n)], axis=1)
for it in range(16):
    Z, s, sw = pj.block_jacobi_rows2(Z, np.arange(n), 8, tol=1e-13, maxsweeps=1)
    Y, W = Z[:, :n], Z[:, n:]
    G = Y@Y.conj().T; d=np.sqrt(np.diag(G).real); off=np.abs(G)/np.outer(d,d); np.fill_diagonal(off,0)
    i,j = np.unravel_index(off.argmax(), off.shape)
    print(it, "inv", np.abs(Y-W@C).max(), "unit", np.abs(W@W.conj().T-np.eye(n)).max(), "top s", s[:3], "worst", (i,j), off[i,j], d[i], d[j])
rows = np.r_[np.arange(0,8), np.arange(40,48)]
Xp = Z[rows][:, :n]
G = Xp@Xp.conj().T
Q, w = pj.herm_jacobi_evd(G)
D = Q.conj().T@G@Q
d=np.sqrt(np.abs(np.diag(D))); off=np.abs(D)/np.outer(d,d); np.fill_diagonal(off,0)
print("inner: rel off of Q^H G Q", off.max(), "w", w)
Xn = Q.conj().T@Xp
G2 = Xn@Xn.conj().T
d=np.sqrt(np.abs(np.diag(G2))); off=np.abs(G2)/np.outer(d,d); np.fill_diagonal(off,0)
print("after applying to rows: rel off", off.max(), "norms", d)
