"""Bulge chase BY BAND POSITION — index-logic prototype (numpy, sequential).

Position j owns the blocks E_j = A[R_j, R_{j-1}] and D_j = A[R_j, R_j] (R_j(s) = s + 1 + j SB + [0, SB)) for ALL sweeps s.
The blocks live in "slot" coordinates: element (r, c) at [r & 31][c & 31]; when the window slides by one row and column per
sweep the slot of the dropped first row / column (o_old = r0_old & 31) receives the new last row / column.  What crosses
between positions per task: the reflector v_{j-1}(s) (left -> right), and from the right neighbour the first row of
E_{j+1}(s-1) + D_{j+1}(s-1)[0, 0].  Checked against the plain sweep-by-sweep algorithm on the full matrix (d, e, v, tau
equal to rounding) for sizes that exercise short last blocks."""
import sys

import numpy as np

SB = 32


def reflector(xn2, alpha):
    if (xn2 == 0.0 and alpha.imag == 0.0):
        return 0.0 + 0.0j, alpha.real, 0.0 + 0.0j
    beta = -np.copysign(np.sqrt(alpha.real ** 2 + alpha.imag ** 2 + xn2), alpha.real)
    tau = complex((beta - alpha.real) / beta, -alpha.imag / beta)
    scal = 1.0 / (alpha - beta)
    return tau, beta, scal


def chase_reference(A):
    """Plain algorithm on the full Hermitian matrix (the task order of sb_chase_sweep)."""
    A = A.copy()
    n = A.shape[0]
    d, e = np.zeros(n), np.zeros(n)
    refl = {}
    for s in range(n - 1):
        d[s] = A[s, s].real
        vprev = tauprev = None
        j = 0
        while True:
            r0 = s + 1 + j * SB
            if r0 >= n:
                break
            nr = min(SB, n - r0)
            R = slice(r0, r0 + nr)
            if j == 0:
                x = A[R, s].copy()
                tau, beta, scal = reflector(float(np.sum(np.abs(x[1:]) ** 2)), x[0])
                v = x * scal
                v[0] = 1.0
                A[R, s] = 0.0
                A[r0, s] = beta
                A[s, R] = 0.0
                A[s, r0] = beta
                e[s] = beta
            else:
                C = slice(r0 - SB, r0)
                E = A[R, C]
                E = E - np.outer(tauprev * (E @ vprev), vprev.conj())
                if nr < 2:
                    A[R, C] = E
                    A[C, R] = E.conj().T
                    break
                x = E[:, 0].copy()
                tau, beta, scal = reflector(float(np.sum(np.abs(x[1:]) ** 2)), x[0])
                v = x * scal
                v[0] = 1.0
                y = np.conj(tau) * (v.conj() @ E)
                E = E - np.outer(v, y)
                E[:, 0] = 0.0
                E[0, 0] = beta
                A[R, C] = E
                A[C, R] = E.conj().T
            D = A[R, R]
            xx = tau * (D @ v)
            al = -0.5 * tau * (xx.conj() @ v)
            w = xx + al * v
            D = D - np.outer(v, w.conj()) - np.outer(w, v.conj())
            A[R, R] = D
            vfull = np.zeros(SB, complex)
            vfull[:nr] = v
            refl[(s, j)] = (vfull, tau)
            vprev, tauprev = vfull, tau
            j += 1
    d[n - 1] = A[n - 1, n - 1].real
    return d, e, refl


class Pos(object):
    def __init__(self, j, A):
        """Blocks of position j at sweep 0 from the band (rows >= n are zero)."""
        n = A.shape[0]
        self.j, self.n = j, n
        self.E = np.zeros((SB, SB), complex)
        self.D = np.zeros((SB, SB), complex)
        r0 = 1 + j * SB
        for r in range(r0, min(r0 + SB, n)):
            for c in range(r0 - SB, r0):
                if c >= 0:
                    self.E[r & 31, c & 31] = A[r, c]
            for c in range(r0, min(r0 + SB, n)):
                self.D[r & 31, c & 31] = A[r, c]
        # mailboxes written by this position
        self.V = None        # (sweep, v by slot, tau)
        self.ROW = {}        # sweep -> first row of E by column slot (corner = beta)
        self.DCOL = None     # (sweep, first column of D by row slot)
        self.DCORN = {}      # sweep -> D[0, 0]


def chase_by_position(A):
    n = A.shape[0]
    jb = (n - 2) // SB + 1
    P = [Pos(j, A) for j in range(jb)]
    d, e = np.zeros(n), np.zeros(n)
    d[0] = A[0, 0].real
    refl = {}

    def active(j, s):
        return j < jb and s >= 0 and s + 1 + j * SB < n

    for s in range(n - 1):
        for j in range(jb):
            r0 = s + 1 + j * SB
            if r0 >= n:
                break
            p = P[j]
            nr = min(SB, n - r0)
            o = r0 & 31
            oo = (r0 - 1) & 31       # slot of the dropped first row / column = slot of the new last ones
            right = active(j + 1, s - 1)
            # ---------------- E wave
            if s > 0:
                assert p.DCOL[0] == s - 1
                col = p.DCOL[1].copy()           # D_j(s-1)[:, first col] by row slot; its slot oo entry is not used
                corner = P[j + 1].ROW[s - 1][oo] if right else 0.0
                p.E[oo, :] = 0.0                 # new last row: zeros ...
                p.E[:, oo] = col
                p.E[oo, oo] = corner             # ... except the corner
            if j == 0:
                x = p.E[:, (r0 - 1) & 31] if s > 0 else p.E[:, 0 & 31]   # column s (slot s & 31 = (r0 - 1) & 31)
                x = x.copy()
                rest = [k for k in range(SB) if k != o]
                tau, beta, scal = reflector(float(np.sum(np.abs(x[rest]) ** 2)), x[o])
                v = x * scal
                v[o] = 1.0
                e[s] = beta
                reflect = True
            else:
                assert P[j - 1].V[0] == s
                vp, taup = P[j - 1].V[1], P[j - 1].V[2]
                p.E = p.E - np.outer(taup * (p.E @ vp), vp.conj())
                reflect = nr >= 2
                if reflect:
                    x = p.E[:, o].copy()
                    rest = [k for k in range(SB) if k != o]
                    tau, beta, scal = reflector(float(np.sum(np.abs(x[rest]) ** 2)), x[o])
                    v = x * scal
                    v[o] = 1.0
                    y = np.conj(tau) * (v.conj() @ p.E)
                    p.E = p.E - np.outer(v, y)
                    p.E[o, o] = beta
                p.ROW[s] = p.E[o, :].copy()
            if reflect:
                p.V = (s, v.copy(), tau)
                vfull = np.array([v[(r0 + k) & 31] for k in range(SB)])
                refl[(s, j)] = (vfull, tau)
            # ---------------- D wave
            if s > 0:
                if right:
                    row = P[j + 1].ROW[s - 1].copy()
                    dc = P[j + 1].DCORN[s - 1]
                else:
                    row, dc = np.zeros(SB, complex), 0.0
                row[oo] = 0.0                    # (the corner of the packet belongs to E)
                p.D[oo, :] = row
                p.D[:, oo] = row.conj()
                p.D[oo, oo] = dc
            if reflect:
                xx = tau * (p.D @ v)
                al = -0.5 * tau * (xx.conj() @ v)
                w = xx + al * v
                p.D = p.D - np.outer(v, w.conj()) - np.outer(w, v.conj())
                p.DCOL = (s, p.D[:, o].copy())
            p.DCORN[s] = p.D[o, o].real
            if j == 0:
                d[s + 1] = p.D[o, o].real
            # ring depth check: the prototype keeps dictionaries; the kernel keeps two slots
            for k in list(p.ROW):
                if k < s - 1:
                    del p.ROW[k]
            for k in list(p.DCORN):
                if k < s - 1:
                    del p.DCORN[k]
    return d, e, refl


def main():
    rng = np.random.default_rng(3)
    ok = True
    for n in [int(a) for a in sys.argv[1:]] or [34, 35, 40, 64, 65, 66, 67, 97, 100, 129, 161]:
        B = rng.standard_normal((n, n)) + 1j * rng.standard_normal((n, n))
        B = B + B.conj().T
        A = np.zeros((n, n), complex)
        for k in range(-SB, SB + 1):
            A += np.diag(np.diag(B, k), k)
        d0, e0, r0 = chase_reference(A)
        d1, e1, r1 = chase_by_position(A)
        T = np.diag(d0) + np.diag(e0[: n - 1], 1) + np.diag(e0[: n - 1], -1)
        ev_ok = np.abs(np.linalg.eigvalsh(T) - np.linalg.eigvalsh(A)).max()
        errd, erre = np.abs(d0 - d1).max(), np.abs(e0 - e1).max()
        same_keys = set(r0) == set(r1)
        errv = max(np.abs(r0[k][0] - r1[k][0]).max() for k in r0) if same_keys else np.inf
        errt = max(abs(r0[k][1] - r1[k][1]) for k in r0) if same_keys else np.inf
        print("n %4d  reference tridiagonal vs eig(A) %.1e   by-position: d %.1e e %.1e v %.1e tau %.1e keys %s"
              % (n, ev_ok, errd, erre, errv, errt, same_keys))
        ok = ok and ev_ok < 1e-11 * n and max(errd, erre, errv, errt) < 1e-11 and same_keys
    print("OK" if ok else "FAILED")
    return 0 if ok else 1


if __name__ == "__main__":
    sys.exit(main())
