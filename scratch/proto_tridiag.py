"""numpy prototype of the tridiagonalisation-based Hermitian eigensolver planned for the GPU:
  T1 blocked Householder tridiagonalisation (panel of nb reflectors, lazily applied rank-2k update)
  T2 implicit QL on the real tridiagonal, recording the Givens rotations sweep by sweep
  T3 rotations applied to Z = I (real)
  T4 back-transformation with compact-WY blocks
"""
import numpy as np, scipy.linalg as la

def tridiag_blocked(A, nb=8):
    A = A.copy(); n = A.shape[0]
    d = np.zeros(n); e = np.zeros(max(n-1,0)); tau = np.zeros(max(n-1,0), dtype=complex)
    Vt = np.zeros((n, n), dtype=complex)   # row k = v_k (v_k[j] nonzero for j >= k+1, v_k[k+1] = 1)
    for k0 in range(0, n-1, nb):
        k1 = min(k0+nb, n-1)
        V = np.zeros((n, k1-k0), dtype=complex); W = np.zeros((n, k1-k0), dtype=complex)
        for k in range(k0, k1):
            j = k-k0
            # column k of the lazily-updated matrix
            col = A[:, k] - V[:, :j] @ W[k, :j].conj() - W[:, :j] @ V[k, :j].conj()
            d[k] = col[k].real
            x = col[k+1:].copy()
            alpha = x[0]; xnorm = np.linalg.norm(x[1:])
            if xnorm == 0.0 and alpha.imag == 0.0:
                t = 0.0; beta = alpha.real; v = np.zeros(n-k-1, dtype=complex); v[0] = 1.0
            else:
                beta = -np.copysign(np.sqrt(abs(alpha)**2 + xnorm**2), alpha.real)
                t = complex((beta-alpha.real)/beta, -alpha.imag/beta)
                v = x/(alpha-beta); v[0] = 1.0
            e[k] = beta; tau[k] = t
            vf = np.zeros(n, dtype=complex); vf[k+1:] = v
            Vt[k] = vf
            # p = tau * (A_lazy v)
            p = A @ vf - V[:, :j] @ (W[:, :j].conj().T @ vf) - W[:, :j] @ (V[:, :j].conj().T @ vf)
            p[:k+1] = 0
            p *= t
            w = p - 0.5*t*np.vdot(p, vf)*vf       # zhetd2: alpha = -half*tau*zdotc(x, v); w = x + alpha v
            V[:, j] = vf; W[:, j] = w
        A -= V @ W.conj().T + W @ V.conj().T
    d[n-1] = A[n-1, n-1].real
    return d, e, tau, Vt

def ql_implicit(d, e, maxit=60):
    """QL with implicit Wilkinson shifts (EISPACK tql2 / NR tqli); records rotations.
    Returns eigenvalues (unsorted) and a list of sweeps (l, m, cs) where cs[i-l] = (c, s) for plane (i, i+1),
    applied for i = m-1 down to l to the columns of Z."""
    d = d.copy(); n = d.size
    e = np.concatenate([e.copy(), [0.0]])
    sweeps = []
    for l in range(n):
        it = 0
        while True:
            m = l
            while m < n-1:
                dd = abs(d[m]) + abs(d[m+1])
                if abs(e[m]) <= np.finfo(float).eps * dd: break
                m += 1
            if m == l: break
            it += 1
            if it > maxit: raise RuntimeError("no convergence")
            g = (d[l+1]-d[l])/(2.0*e[l]); r = np.hypot(g, 1.0)
            g = d[m]-d[l]+e[l]/(g+np.copysign(r, g))
            s = c = 1.0; p = 0.0
            cs = np.zeros((m-l, 2)); cs[:, 0] = 1.0
            i = m-1
            underflow = False
            while i >= l:
                f = s*e[i]; b = c*e[i]
                r = np.hypot(f, g); e[i+1] = r
                if r == 0.0:
                    d[i+1] -= p; e[m] = 0.0; underflow = True; break
                s = f/r; c = g/r
                g = d[i+1]-p
                r = (d[i]-g)*s + 2.0*c*b
                p = s*r; d[i+1] = g+p; g = c*r-b
                cs[i-l] = (c, s)
                i -= 1
            sweeps.append((l, m, cs, i+1))   # rotations valid for planes i+1..m-1
            if underflow: continue
            d[l] -= p; e[l] = g; e[m] = 0.0
    return d, sweeps

def apply_sweeps(Z, sweeps):
    for (l, m, cs, ilo) in sweeps:
        for i in range(m-1, ilo-1, -1):
            c, s = cs[i-l]
            zi1 = Z[:, i+1].copy(); zi = Z[:, i].copy()
            Z[:, i+1] = s*zi + c*zi1
            Z[:, i] = c*zi - s*zi1
    return Z

def back_transform(Vt, tau, Z, nb=8):
    n = Z.shape[0]; X = Z.astype(complex)
    nref = n-1
    blocks = [(k0, min(k0+nb, nref)) for k0 in range(0, nref, nb)]
    for (k0, k1) in reversed(blocks):
        Vb = Vt[k0:k1].T                      # n x kb
        kb = k1-k0
        T = np.zeros((kb, kb), dtype=complex)
        G = Vb.conj().T @ Vb
        for j in range(kb):
            T[j, j] = tau[k0+j]
            if j > 0:
                T[:j, j] = -tau[k0+j] * (T[:j, :j] @ G[:j, j])
        X -= Vb @ (T @ (Vb.conj().T @ X))
    return X

def herm_eig_tridiag(C, nb=8):
    n = C.shape[0]
    d, e, tau, Vt = tridiag_blocked(C, nb)
    lam, sweeps = ql_implicit(d, e)
    Z = apply_sweeps(np.eye(n), sweeps)
    X = back_transform(Vt, tau, Z, nb)
    return lam, X, len(sweeps), sum(s[1]-s[3] for s in sweeps)

if __name__ == "__main__":
    rng = np.random.default_rng(0)
    for n in (5, 40, 150):
        X = rng.standard_normal((n,n)) + 1j*rng.standard_normal((n,n))
        lam0 = 10.0**rng.uniform(-12, 0, n)
        Q = np.linalg.qr(X)[0]
        C = (Q*lam0)@Q.conj().T; C = 0.5*(C+C.conj().T)
        d, e, tau, Vt = tridiag_blocked(C, 8)
        T = np.diag(d) + np.diag(e, 1) + np.diag(e, -1)
        print("n", n, "tridiag eig err", np.abs(np.linalg.eigvalsh(T) - np.linalg.eigvalsh(C)).max())
        lam, Xv, nsw, nrot = herm_eig_tridiag(C)
        o = np.argsort(lam); ref = np.linalg.eigvalsh(C)
        print("   sweeps", nsw, "rot/n^2", nrot/n**2, "eval err", np.abs(lam[o]-ref).max(), "resid", np.abs(C@Xv - Xv*lam).max(), "unit", np.abs(Xv.conj().T@Xv-np.eye(n)).max())
    g = np.load("/root/repo/tests/golden/svdkl_unpol_harsh.npz")
    cs, cn = g["m5_kl_cs"], g["m5_kl_cn"]
    L = la.cholesky(cn, lower=True); Xs = la.solve_triangular(L, cs, lower=True)
    C = la.solve_triangular(L, Xs.conj().T, lower=True); C = 0.5*(C+C.conj().T)
    lam, Xv, nsw, nrot = herm_eig_tridiag(C)
    ref = np.linalg.eigvalsh(C)
    print("harsh KL: eval err/scale", np.abs(np.sort(lam)-ref).max()/ref.max(), "rel err of small", np.abs(np.sort(lam)[:5]/ref[:5]-1))
