import numpy as np, scipy.linalg as la, sys
sys.path.insert(0,'/root/repo'); sys.path.insert(0,'/root/repo/scratch')
from proto_jacobi import block_jacobi_rows

def eigh_gen_jacobi(A, B, b=8, verbose=False):
    n = A.shape[0]
    L = la.cholesky(B, lower=True)
    X = la.solve_triangular(L, A, lower=True)
    C = la.solve_triangular(L, X.conj().T, lower=True).conj().T
    C = 0.5*(C + C.conj().T)
    Z = np.concatenate([C, np.eye(n)], axis=1)
    Z, s, sw = block_jacobi_rows(Z, np.arange(n), b, verbose=verbose)
    Y, W = Z[:, :n], Z[:, n:]      # Y = W C ; rows of W are u_i^H
    lam = np.real(np.sum(Y * W.conj(), axis=1))   # u^H C u
    order = np.argsort(lam, kind="stable")
    lam = lam[order]; W = W[order]
    # E = W L^-1  (rows are modes): E^H = L^-H W^H
    E = la.solve_triangular(L.conj().T, W.conj().T, lower=False).conj().T
    return lam, E, sw

g = np.load("/root/repo/tests/golden/svdkl_unpol.npz")
for m in g["mlist"]:
    for name in ("kl","klnf"):
        cs, cn = g[f"m{m}_{name}_cs"], g[f"m{m}_{name}_cn"]
        lam, E, sw = eigh_gen_jacobi(cs, cn)
        ref = g[f"m{m}_{name}_evals"]
        print(name, m, "n", len(lam), "sweeps", sw, "eval err/scale", np.abs(lam-ref).max()/np.abs(ref).max(), "ENE-I", np.abs(E@cn@E.conj().T-np.eye(len(lam))).max())

# bigger KL-like problem
rng = np.random.default_rng(0)
n = 320
Bs = rng.standard_normal((n, 200)) + 1j*rng.standard_normal((n,200))
S = (Bs * (1.0/(1+np.arange(200))**2)) @ Bs.conj().T * 1e-3
Bf = rng.standard_normal((n, 12)) + 1j*rng.standard_normal((n,12))
N = (Bf*1e6) @ Bf.conj().T + np.eye(n)
ref = la.eigh(S, N, eigvals_only=True)
for b in (8, 16, 32):
    lam, E, sw = eigh_gen_jacobi(S, N, b=b, verbose=False)
    print("n", n, "b", b, "sweeps", sw, "err/scale", np.abs(lam-ref).max()/np.abs(ref).max(), "rel err top10", np.abs(lam[-10:]/ref[-10:]-1).max())
