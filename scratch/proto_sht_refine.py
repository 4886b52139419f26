"""Prototype: healpy-style Jacobi refinement of map2alm carried out in harmonic space (no maps).
a_{k+1} = a_0 + a_k - A(S(a_k));  A o S = K (per-m Gram matrix of the ring functions under the quadrature)
+ alias terms on the polar rings with N_r <= 2 lmax.  Checked against oracle.btgen.transfer_single(niter=...)."""
import sys, time
import numpy as np
sys.path.insert(0, "/root/repo")
from oracle import btgen as ob


def refine_harmonic(a0, nside, lmax, polarised, niter, cull=None):
    """a0: dict m -> (P, lmax+1-|m|) as ob._analysis returns, all m in -lmax..lmax."""
    z, nphi, phi0, start = ob.ring_info(nside)
    npix = 12 * nside**2
    w = 4 * np.pi / npix
    P = 4 if polarised else 1
    nring = z.size
    tabs = {}
    for am in range(lmax + 1):
        lam = ob.lambda_lm(lmax, am, z)
        if polarised:
            W, X = ob.wx_lm(lmax, am, z)
        else:
            W = X = None
        tabs[am] = (lam, W, X)

    def synth_ring(c, m):   # F_m[r] for all rings: (P, nring)
        am = abs(m); lam, W, X = tabs[am]
        sgn = (-1.0) ** am if m < 0 else 1.0
        F = np.zeros((P, nring), complex)
        F[0] = sgn * (c[0] @ lam)
        if polarised:
            sx = -sgn if m < 0 else 1.0
            F[1] = sgn * (c[1] @ W) - 1j * sx * (c[2] @ X)
            F[2] = sgn * (c[2] @ W) + 1j * sx * (c[1] @ X)
            F[3] = sgn * (c[3] @ lam)
        return F

    def ana_ring(g, m):     # g (nring, P) -> c (P, L-am)
        am = abs(m); lam, W, X = tabs[am]
        sgn = (-1.0) ** am if m < 0 else 1.0
        c = np.zeros((P, lmax + 1 - am), complex)
        c[0] = sgn * ((lam * w) @ g[:, 0])
        if polarised:
            sx = -sgn if m < 0 else 1.0
            c[1] = sgn * ((W * w) @ g[:, 1]) - 1j * sx * ((X * w) @ g[:, 2])
            c[2] = sgn * ((W * w) @ g[:, 2]) + 1j * sx * ((X * w) @ g[:, 1])
            c[3] = sgn * ((lam * w) @ g[:, 3])
        return c

    a = {m: c.copy() for m, c in a0.items()}
    for _ in range(niter):
        F = {m: synth_ring(a[m], m) for m in a}     # (P, nring)
        new = {}
        for mp in a:
            G = np.zeros((nring, P), complex)
            for r in range(nring):
                N = int(nphi[r])
                # all m = mp - k N within the band
                kmin = int(np.ceil((mp - lmax) / N)); kmax = int(np.floor((mp + lmax) / N))
                for k in range(kmin, kmax + 1):
                    m = mp - k * N
                    ph = np.exp(1j * (mp - m) * phi0[r])
                    if cull is not None and k != 0 and (abs(m) > cull[r] or abs(mp) > cull[r]):
                        continue
                    G[r] += N * ph * F[m][:, r]
            new[mp] = a0[mp] + a[mp] - ana_ring(G, mp)
        a = new
    return a


def mlim_table(nside, lmax, polarised, eps):
    z, nphi, phi0, start = ob.ring_info(nside)
    nring = z.size
    big = np.zeros((lmax + 1, nring))
    for m in range(lmax + 1):
        lam = ob.lambda_lm(lmax, m, z)
        v = np.abs(lam).max(axis=0)
        if polarised:
            W, X = ob.wx_lm(lmax, m, z)
            v = np.maximum(v, np.maximum(np.abs(W).max(axis=0), np.abs(X).max(axis=0)))
        big[m] = v
    # mlim[r] = largest m with big[m, r] >= eps
    ml = np.zeros(nring, dtype=int)
    for r in range(nring):
        nz = np.nonzero(big[:, r] >= eps)[0]
        ml[r] = nz.max() if nz.size else -1
    return ml


if __name__ == "__main__":
    rng = np.random.default_rng(1)
    for polarised in (False, True):
        nside, lmax = 8, 14
        P = 4 if polarised else 1
        npix = 12 * nside**2
        maps = rng.standard_normal((P, npix)) + 1j * rng.standard_normal((P, npix))
        ms = np.arange(-lmax, lmax + 1)
        a0 = ob._analysis(maps, nside, lmax, polarised, ms)
        for niter in (1, 3):
            ref = ob.transfer_single(maps if polarised else maps[0], nside, lmax, lmax, polarised, niter=niter)
            a = refine_harmonic(a0, nside, lmax, polarised, niter)
            err = 0
            for m, c in a.items():
                col = m if m >= 0 else 2 * lmax + 1 + m
                err = max(err, np.abs(ref[:, abs(m):, col] - c).max())
            print("pol", polarised, "niter", niter, "max err", err, "scale", np.abs(ref).max())
    # cull limits at realistic sizes
    for nside, lmax in ((128, 96), (128, 128), (64, 96), (512, 512), (256, 383)):
        for pol in (False, True):
            t0 = time.time()
            ml = mlim_table(nside, lmax, pol, 1e-22)
            z, nphi, phi0, start = ob.ring_info(nside)
            alias = [r for r in range(nside - 1) if 2 * min(ml[r], lmax) >= nphi[r]]
            print(nside, lmax, pol, "alias rings (north):", len(alias), "last", alias[-1] if alias else None,
                  "Mcut", max(min(ml[r], lmax) for r in alias) if alias else None, "t %.1fs" % (time.time() - t0))
