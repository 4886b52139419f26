"""numpy prototype of the divide & conquer merge (Cuppen / Gu-Eisenstat, LAPACK dlaed2/3/4 scheme)
planned for the GPU tridiagonal eigensolver."""
import numpy as np

EPS = np.finfo(float).eps

def secular_root(j, d, z2, rho, maxit=60):
    """Root j of 1 + rho sum z2_i/(d_i - lam) = 0 in (d_j, d_{j+1}) (last: (d_k-1, d_k-1 + rho sum z2)).
    Returns (origin index o, mu) with lam = d[o] + mu; delta_i = (d_i - d[o]) - mu accurate."""
    k = d.size
    if k == 1:
        return 0, rho * z2[0]
    last = (j == k - 1)
    if not last:
        gap = d[j + 1] - d[j]
        mid = 0.5 * gap
        dl = d - d[j]
        fmid = 1.0 + rho * np.sum(z2 / (dl - mid))
        if fmid > 0:      # root in the left half: origin d_j, mu in (0, mid]
            o = j; lo, hi = 0.0, mid
        else:             # origin d_{j+1}, mu in [-mid, 0)
            o = j + 1; lo, hi = -mid, 0.0
    else:
        o = j
        lo, hi = 0.0, rho * np.sum(z2)
        # f(hi) >= 0 always
    dl = d - d[o]
    # safeguarded rational iteration on mu: bracket [lo, hi], f increasing
    def f_parts(mu):
        t = dl - mu
        terms = z2 / t
        # psi: poles <= root side (i <= j), phi: poles > j
        psi = rho * np.sum(terms[: j + 1]); phi = rho * np.sum(terms[j + 1:])
        dpsi = rho * np.sum(z2[: j + 1] / t[: j + 1] ** 2); dphi = rho * np.sum(z2[j + 1:] / t[j + 1:] ** 2)
        return 1.0 + psi + phi, psi, phi, dpsi, dphi, np.abs(terms).sum() * rho
    mu = 0.5 * (lo + hi)
    for it in range(maxit):
        fv, psi, phi, dpsi, dphi, mag = f_parts(mu)
        erretm = 8.0 * (abs(psi) + abs(phi)) + 1.0 + abs(mu) * (dpsi + dphi)
        if abs(fv) <= EPS * erretm:
            break
        if fv > 0: hi = mu
        else: lo = mu
        # "middle way" rational model: psi ~ s + p/(dj - mu'), phi ~ r + q/(dj1 - mu')
        if not last:
            dj = dl[j] - mu; dj1 = dl[j + 1] - mu      # distances to the two neighbouring poles
            a = (dj + dj1) * fv - dj * dj1 * (dpsi + dphi)
            b = dj * dj1 * fv
            c = fv - dj * dpsi - dj1 * dphi
            if c == 0:
                eta = b / a if a != 0 else 0.0
            else:
                disc = a * a - 4 * b * c
                disc = max(disc, 0.0)
                eta = (a - np.sqrt(disc)) / (2 * c) if a <= 0 else 2 * b / (a + np.sqrt(disc))
        else:
            # last root (dlaed4, I = N): rational model on the two last poles, both left of the root
            tq = dl[j] - mu; tp = dl[j - 1] - mu                    # delta(n), delta(n-1) < 0
            dphi_l = rho * z2[j] / tq ** 2                           # last pole alone
            dpsi_l = dpsi + dphi - dphi_l                            # all the others
            c = fv - tp * dpsi_l - tq * dphi_l
            a = (tp + tq) * fv - tp * tq * (dpsi_l + dphi_l)
            b = tp * tq * fv
            if c < 0: c = abs(c)
            if c == 0:
                eta = hi - mu
            elif a >= 0:
                eta = (a + np.sqrt(abs(a * a - 4 * b * c))) / (2 * c)
            else:
                eta = 2 * b / (a - np.sqrt(abs(a * a - 4 * b * c)))
            if fv * eta > 0: eta = -fv / (dpsi + dphi)
        new = mu + eta
        if not (lo < new < hi) or not np.isfinite(new):
            new = 0.5 * (lo + hi)
        if new == mu or (hi - lo) <= 2 * EPS * abs(new): 
            mu = new; break
        mu = new
    return o, mu

def merge(d1, Q1, d2, Q2, beta):
    """Eigen-decomposition of diag(T1hat, T2hat) + rho v v^T given the halves' decompositions."""
    n1, n2 = d1.size, d2.size
    n = n1 + n2
    rho = abs(beta)
    z = np.concatenate([Q1[-1, :], np.sign(beta) * Q2[0, :]])
    d = np.concatenate([d1, d2])
    Q = np.zeros((n, n)); Q[:n1, :n1] = Q1; Q[n1:, n1:] = Q2
    # normalise z
    zn = np.linalg.norm(z); z = z / zn; rho = rho * zn * zn
    order = np.argsort(d, kind="stable")
    d = d[order]; z = z[order]; Q = Q[:, order]
    tol = 8 * EPS * max(np.abs(d).max(), np.abs(z).max())
    keep = []
    defl = []
    # deflation pass (dlaed2): tiny z, then close poles via Givens
    prev = -1
    if rho * np.abs(z).max() <= tol:
        return d, Q
    for i in range(n):
        if rho * abs(z[i]) <= tol:
            defl.append(i); continue
        if prev >= 0:
            s = z[prev]; c = z[i]
            tau = np.hypot(c, s)
            t = d[i] - d[prev]
            c /= tau; s = -s / tau
            if abs(t * c * s) <= tol:
                # rotate: z[prev] -> 0
                z[i] = tau; z[prev] = 0.0
                qp = Q[:, prev].copy(); qi = Q[:, i].copy()
                Q[:, prev] = c * qp + s * qi
                Q[:, i] = -s * qp + c * qi
                dp, di = d[prev], d[i]
                d[prev] = dp * c * c + di * s * s
                d[i] = dp * s * s + di * c * c
                defl.append(prev)
                prev = i
                keep[-1] = i
                continue
        keep.append(i); prev = i
    keep = np.array(keep, dtype=int); k = keep.size
    lam_out = d.copy(); Qout = Q.copy()
    if k == 0:
        return lam_out, Qout
    dk = d[keep]; zk = z[keep]
    # poles must be increasing (rotations may have perturbed the order slightly): sort
    o2 = np.argsort(dk, kind="stable"); dk = dk[o2]; zk = zk[o2]; keep = keep[o2]
    z2 = zk * zk
    org = np.zeros(k, dtype=int); mu = np.zeros(k)
    for j in range(k):
        org[j], mu[j] = secular_root(j, dk, z2, rho)
    lam = dk[org] + mu
    # delta[i, j] = d_i - lam_j computed stably
    delta = (dk[:, None] - dk[org][None, :]) - mu[None, :]
    # Gu-Eisenstat: zhat_i^2 = prod_j (lam_j - d_i) / (rho prod_{j != i} (d_j - d_i))
    zhat = np.zeros(k)
    for i in range(k):
        num = -delta[i, :]                      # lam_j - d_i
        den = np.delete(dk - dk[i], i)
        # interleave products to avoid over/underflow
        p = num[i] if k > 0 else 1.0
        idx = [j for j in range(k) if j != i]
        prod = num[i]
        for j, dj in zip(idx, den):
            prod *= num[j] / dj
        zhat[i] = np.sqrt(abs(prod) / rho) * (1.0 if zk[i] >= 0 else -1.0)
    U = zhat[:, None] / delta
    U /= np.linalg.norm(U, axis=0)[None, :]
    Qk = Q[:, keep] @ U
    Qout[:, keep] = Qk
    lam_out[keep] = lam
    return lam_out, Qout

def dc_eig(d, e, leaf=8):
    n = d.size
    if n <= leaf:
        T = np.diag(d) + np.diag(e, 1) + np.diag(e, -1)
        w, V = np.linalg.eigh(T)
        return w, V
    k = n // 2
    beta = e[k - 1]
    d1 = d[:k].copy(); d2 = d[k:].copy()
    d1[-1] -= abs(beta); d2[0] -= abs(beta)
    w1, Q1 = dc_eig(d1, e[: k - 1], leaf)
    w2, Q2 = dc_eig(d2, e[k:], leaf)
    return merge(w1, Q1, w2, Q2, beta)

if __name__ == "__main__":
    rng = np.random.default_rng(0)
    def check(name, d, e):
        n = d.size
        T = np.diag(d) + np.diag(e, 1) + np.diag(e, -1)
        w, Q = dc_eig(d, e)
        ref = np.linalg.eigvalsh(T)
        sc = max(np.abs(ref).max(), 1e-300)
        print(f"{name:18s} n={n:4d} eval err {np.abs(np.sort(w)-ref).max()/sc:.2e}  orth {np.abs(Q.T@Q-np.eye(n)).max():.2e}  resid {np.abs(T@Q-Q*w).max()/sc:.2e}")
    for n in (9, 33, 100, 257):
        check("random", rng.standard_normal(n), rng.standard_normal(n - 1))
    n = 200
    check("graded", 10.0 ** (-np.arange(n) / 12.0), 10.0 ** (-np.arange(n - 1) / 12.0) * 0.3)
    check("clustered", np.ones(n) + 1e-13 * rng.standard_normal(n), 1e-10 * rng.standard_normal(n - 1))
    m = 21; dw = np.abs(np.arange(-10, 11)).astype(float)
    check("wilkinson glued", np.tile(dw, 5), np.concatenate([np.r_[np.ones(m - 1), 1e-10] for _ in range(5)])[:-1])
    check("zeros offdiag", rng.standard_normal(n), np.where(rng.uniform(size=n - 1) < 0.3, 0.0, rng.standard_normal(n - 1)))
    # tridiagonal from a KL-like matrix
    import sys; sys.path.insert(0, "/root/repo/scratch")
    from proto_tridiag import tridiag_blocked
    X = rng.standard_normal((150, 150)) + 1j * rng.standard_normal((150, 150))
    Qm = np.linalg.qr(X)[0]
    C = (Qm * 10.0 ** rng.uniform(-12, 0, 150)) @ Qm.conj().T; C = 0.5 * (C + C.conj().T)
    d, e, tau, Vt = tridiag_blocked(C, 8)
    check("KL-like", d, e)
