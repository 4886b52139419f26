"""Two-sided block Jacobi for Hermitian EVD (numpy prototype)."""
import numpy as np, scipy.linalg as la, sys
sys.path.insert(0,'/root/repo/scratch')
from proto_jacobi import round_robin_pairs
from proto_jacobi2 import herm_jacobi_evd

def herm_jacobi_general(G, tol=1e-15, maxsweeps=30, absfloor=0.0):
    """two-sided cyclic Jacobi for general Hermitian (indefinite ok): abs-scaled threshold."""
    G = G.copy(); k = G.shape[0]; Q = np.eye(k, dtype=complex)
    for sweep in range(maxsweeps):
        rotated = False
        for p in range(k-1):
            for q in range(p+1, k):
                g = G[p,q]; a = G[p,p].real; bq = G[q,q].real; ag = abs(g)
                if ag <= tol*np.sqrt(abs(a*bq)) or ag == 0.0 or ag <= absfloor: continue
                rotated = True
                ph = g/ag; zeta = (bq-a)/(2*ag)
                t = (np.sign(zeta) if zeta != 0 else 1.0)/(abs(zeta)+np.sqrt(1+zeta*zeta))
                c = 1/np.sqrt(1+t*t); s = c*t
                J = np.array([[c, s*ph],[-s*np.conj(ph), c]])
                G[:, [p,q]] = G[:, [p,q]] @ J
                G[[p,q], :] = J.conj().T @ G[[p,q], :]
                Q[:, [p,q]] = Q[:, [p,q]] @ J
        if not rotated: break
    return Q, np.diag(G).real

def block_jacobi_herm(C, b=8, tol=1e-14, maxsweeps=30, sort=True, verbose=False):
    C = C.copy(); n = C.shape[0]; W = np.eye(n, dtype=complex)
    nb = -(-n//b); nb += nb % 2
    rounds = round_robin_pairs(nb)
    absfloor = 2.2e-16*np.abs(np.diag(C)).max()
    for sweep in range(maxsweeps):
        maxoff = 0.0
        for pairs in rounds:
            for (bi,bj) in pairs:
                if bi > bj: bi,bj = bj,bi
                rows = np.r_[np.arange(bi*b,(bi+1)*b), np.arange(bj*b,(bj+1)*b)]; rows = rows[rows<n]
                if rows.size < 2: continue
                G = C[np.ix_(rows,rows)]
                d = np.sqrt(np.abs(np.diag(G).real)); dd = np.outer(d,d); dd[dd==0]=1
                off = np.abs(G-np.diag(np.diag(G)))/dd
                off[np.abs(G) <= absfloor] = 0.0
                mo = off.max(); maxoff = max(maxoff, mo)
                if mo <= tol: continue
                Q, w = herm_jacobi_general(G, absfloor=absfloor)
                if sort:
                    o = np.argsort(-w, kind="stable"); Q = Q[:, o]
                C[rows,:] = Q.conj().T @ C[rows,:]
                C[:,rows] = C[:,rows] @ Q
                W[rows,:] = Q.conj().T @ W[rows,:]
        if verbose: print("sweep", sweep, "maxoff", maxoff)
        if maxoff <= tol: break
    return np.diag(C).real, W, sweep+1

if __name__ == "__main__":
    g = np.load("/root/repo/tests/golden/svdkl_unpol.npz")
    for m in (0,5):
        cs, cn = g[f"m{m}_kl_cs"], g[f"m{m}_kl_cn"]
        L = la.cholesky(cn, lower=True)
        X = la.solve_triangular(L, cs, lower=True)
        C = la.solve_triangular(L, X.conj().T, lower=True).conj().T
        C = 0.5*(C+C.conj().T)
        for sort in (True, False):
            lam, W, sw = block_jacobi_herm(C, 8, sort=sort)
            ref = g[f"m{m}_kl_evals"]
            print("m",m,"sort",sort,"sweeps", sw, "err/scale", np.abs(np.sort(lam)-ref).max()/ref.max(), "rel", np.abs(np.sort(lam)/ref-1).max())
    rng = np.random.default_rng(0)
    n = 320
    Bs = rng.standard_normal((n, 200)) + 1j*rng.standard_normal((n,200))
    S = (Bs * (1.0/(1+np.arange(200))**2)) @ Bs.conj().T * 1e-3
    Bf = rng.standard_normal((n, 12)) + 1j*rng.standard_normal((n,12))
    N = (Bf*1e6) @ Bf.conj().T + np.eye(n)
    ref = la.eigh(S, N, eigvals_only=True)
    L = la.cholesky(N, lower=True)
    X = la.solve_triangular(L, S, lower=True)
    C = la.solve_triangular(L, X.conj().T, lower=True).conj().T
    C = 0.5*(C+C.conj().T)
    for b in (8, 16, 32):
        lam, W, sw = block_jacobi_herm(C, b, tol=1e-13)
        print("n",n,"b",b,"sweeps",sw,"err/scale",np.abs(np.sort(lam)-ref).max()/ref.max())
