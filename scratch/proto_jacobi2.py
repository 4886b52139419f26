"""Block one-sided Jacobi with a two-sided-Jacobi inner solver on the Gram block."""
import numpy as np, scipy.linalg as la, sys
sys.path.insert(0,'/root/repo'); sys.path.insert(0,'/root/repo/scratch')
from proto_jacobi import round_robin_pairs

def herm_jacobi_evd(G, tol=1e-15, maxsweeps=20):
    """Two-sided cyclic Jacobi on Hermitian PSD G. Returns Q with Q^H G Q ~ diagonal."""
    G = G.copy(); k = G.shape[0]; Q = np.eye(k, dtype=complex)
    for sweep in range(maxsweeps):
        rotated = False
        for p in range(k-1):
            for q in range(p+1, k):
                g = G[p,q]; a = G[p,p].real; bq = G[q,q].real
                ag = abs(g)
                if ag <= tol*np.sqrt(abs(a*bq)) or ag == 0.0: continue
                rotated = True
                ph = g/ag
                zeta = (bq-a)/(2*ag)
                t = np.sign(zeta)/(abs(zeta)+np.sqrt(1+zeta*zeta)) if zeta != 0 else 1.0
                c = 1/np.sqrt(1+t*t); s = c*t
                # J = [[c, s*ph],[-s*conj(ph), c]] ; G <- J^H G J
                J = np.array([[c, s*ph],[-s*np.conj(ph), c]])
                G[:, [p,q]] = G[:, [p,q]] @ J
                G[[p,q], :] = J.conj().T @ G[[p,q], :]
                Q[:, [p,q]] = Q[:, [p,q]] @ J
        if not rotated: break
    return Q, np.diag(G).real

SORT=True
FLOOR=1e-10
dmax_global=0.0
def block_jacobi_rows2(Z, gcols, b=8, tol=1e-14, maxsweeps=30, verbose=False):
    Z = Z.copy(); n = Z.shape[0]
    nb = -(-n // b)
    if nb % 2: nb += 1
    rounds = round_robin_pairs(nb) if nb > 1 else [[(0,0)]]
    global dmax_global
    dmax_global = np.linalg.norm(Z[:, gcols], axis=1).max()
    for sweep in range(maxsweeps):
        maxoff = 0.0
        for pairs in rounds:
            for (bi, bj) in pairs:
                if bi > bj: bi, bj = bj, bi
                rows = np.r_[np.arange(bi*b, (bi+1)*b), np.arange(bj*b, (bj+1)*b)] if bi != bj else np.arange(bi*b, (bi+1)*b)
                rows = rows[rows < n]
                if rows.size < 2: continue
                X = Z[rows][:, gcols]
                G = X @ X.conj().T
                d = np.sqrt(np.abs(np.diag(G).real)); dd = np.outer(d,d); dd[dd==0]=1.0
                off = np.abs(G - np.diag(np.diag(G)))/dd
                if FLOOR > 0:
                    dmax = max(dmax_global, d.max())
                    small = d < FLOOR*dmax
                    off[np.ix_(small, small)] = 0.0
                mo = off.max(); maxoff = max(maxoff, mo)
                if mo <= tol: continue
                Q, w = herm_jacobi_evd(G)
                if SORT:
                    order = np.argsort(-w, kind="stable"); Q = Q[:, order]
                Z[rows] = Q.conj().T @ Z[rows]
        if verbose: print("sweep", sweep, "maxoff", maxoff)
        if maxoff <= tol: break
    sig = np.linalg.norm(Z[:, gcols], axis=1)
    order = np.argsort(-sig, kind="stable")
    return Z[order], sig[order], sweep+1

if __name__ == "__main__":
    g = np.load("/root/repo/tests/golden/svdkl_unpol.npz")
    for m in (0, 5):
        cs, cn = g[f"m{m}_kl_cs"], g[f"m{m}_kl_cn"]
        n = cs.shape[0]
        L = la.cholesky(cn, lower=True)
        X = la.solve_triangular(L, cs, lower=True)
        C = la.solve_triangular(L, X.conj().T, lower=True).conj().T
        C = 0.5*(C+C.conj().T)
        Z = np.concatenate([C, np.eye(n)], axis=1)
        Zr, s, sw = block_jacobi_rows2(Z, np.arange(n), 8, verbose=True, tol=1e-13)
        Y, W = Zr[:, :n], Zr[:, n:]
        lam = np.sort(np.real(np.sum(Y*W.conj(), axis=1)))
        ref = g[f"m{m}_kl_evals"]
        print("sweeps", sw, "err/scale", np.abs(lam-ref).max()/ref.max(), "max rel err", np.abs(lam/ref-1).max())
