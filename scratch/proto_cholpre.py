"""Veselic-Hari style: eig(C) for PSD C via Cholesky C+dI = R R^H, then one-sided Jacobi on the
rows of R^H (= columns of R)."""
import numpy as np, scipy.linalg as la, sys, time
sys.path.insert(0,'/root/repo/scratch')
import proto_jacobi2 as pj
pj.SORT=False

def eig_cholpre(C, b=8, shift=1e-14, pivot=False):
    n = C.shape[0]
    scale = np.abs(np.diag(C)).max()
    d = shift*scale
    Cs = C + d*np.eye(n)
    perm = np.arange(n)
    if pivot:
        perm = np.argsort(-np.diag(Cs).real, kind="stable")
        Cs = Cs[np.ix_(perm, perm)]
    R = la.cholesky(Cs, lower=True)
    Z = R.conj().T.copy()
    Zr, s, sw = pj.block_jacobi_rows2(Z, np.arange(n), b, tol=1e-13)
    lam = s**2 - d
    P = (Zr / s[:, None]).conj().T     # columns = eigenvectors of permuted Cs
    Pfull = np.zeros_like(P); Pfull[perm] = P
    return lam, Pfull, sw

rng = np.random.default_rng(0)
for n in (80, 200, 320):
    X = rng.standard_normal((n,n)) + 1j*rng.standard_normal((n,n))
    lamt = 10.0**rng.uniform(-12, 0, n)
    Q = np.linalg.qr(X)[0]
    C = (Q*lamt)@Q.conj().T; C = 0.5*(C+C.conj().T)
    ref = np.linalg.eigvalsh(C)
    for pivot in (False, True):
        t0=time.time()
        lam, P, sw = eig_cholpre(C, b=8, pivot=pivot)
        o = np.argsort(lam)
        print("n", n, "pivot", pivot, "sweeps", sw, "err/scale", np.abs(lam[o]-ref).max()/ref.max(),
              "unit", np.abs(P.conj().T@P-np.eye(n)).max(), "resid", np.abs(C@P - P*lam).max(), round(time.time()-t0,1),"s")
