"""CPU restatement (numpy) of driftscan's beam-transfer generation for cylinder
telescopes: HEALPix geometry, cylinder beams, the pixel kernels of
``_fast_tools.pyx``, the spherical-harmonic quadrature and the +/-m fold.

TEST INFRASTRUCTURE: the checker for the HIP path and the timed ``cpu_baseline``
of bench.py.  Never imported by the product.

Pinning status
  * ``fringe``, ``horizon``, ``beam_exptan``, ``construct_pol_real``, ``beam_amp``,
    ``beam_x/y``, ``fraunhofer_cylinder``: PINNED against the reference's own compiled
    Cython extension and Python (tests/golden/pixel_kernels.npz, and live against
    oracle/_ref when present).
  * ``sht_*`` (map -> a_lm): PARITY UNPINNED.  The reference calls
    cora.util.hputil.sphtrans_complex[_pol] -> healpy.map2alm (libsharp); neither is
    vendored or installed and the reference's golden tarball is network-only
    (SURVEY.md §8c).  Restated here from the published HEALPix algorithm
    (Gorski et al. 2005): RING pixel centres, equal pixel weights 4 pi / npix, no
    Jacobi iterations (``iter=0``), E/B in the HEALPix convention
    a^E = -(a_2 + a_-2)/2, a^B = i (a_2 - a_-2)/2.  Checked against brute-force sums
    over explicitly constructed (spin-weighted) spherical harmonics instead
    (tests/test_oracle_sht.py).

Follows (paths relative to the reference tree):
  * pixel kernels          drift/util/_fast_tools.pyx:18-282
  * horizon                drift/core/visibility.py:27-46
  * cylinder beams         drift/telescope/cylbeam.py:10-212, cylinder.py:171-218
  * per-(baseline,freq) maps   drift/core/telescope.py:1156-1176 (unpol), :1268-1283 (pol)
  * transfer_single        drift/core/telescope.py:1178-1193 (unpol), :1287-1316 (pol)
  * transfer_matrices      drift/core/telescope.py:755-830
  * fold to m-order        drift/core/beamtransfer.py:620-624, :663
"""
import numpy as np


# ----------------------------------------------------------------------------
# HEALPix RING geometry
# ----------------------------------------------------------------------------
def nside_for_lmax(lmax, accuracy_boost=1):
    """cora.util.hputil.nside_for_lmax (restated)."""
    return int(2 ** (accuracy_boost + np.ceil(np.log((lmax + 1) / 3.0) / np.log(2.0))))


def ring_info(nside):
    """Per-ring z = cos(theta), number of pixels, phi of the first pixel, first pixel index."""
    nring = 4 * nside - 1
    i = np.arange(1, nring + 1)
    z = np.empty(nring)
    nphi = np.empty(nring, dtype=np.int64)
    phi0 = np.empty(nring)
    north = i < nside
    south = i > 3 * nside
    eq = ~(north | south)
    z[north] = 1.0 - i[north] ** 2 / (3.0 * nside**2)
    nphi[north] = 4 * i[north]
    phi0[north] = np.pi / (4.0 * i[north])
    ie = i[eq]
    z[eq] = (2.0 * nside - ie) * 2.0 / (3.0 * nside)
    nphi[eq] = 4 * nside
    shifted = ((ie + nside) % 2) == 0
    phi0[eq] = np.where(shifted, np.pi / (4.0 * nside), 0.0)
    js = 4 * nside - i[south]
    z[south] = -(1.0 - js**2 / (3.0 * nside**2))
    nphi[south] = 4 * js
    phi0[south] = np.pi / (4.0 * js)
    start = np.concatenate([[0], np.cumsum(nphi)[:-1]])
    return z, nphi, phi0, start


def ang_positions(nside):
    """(npix, 2) array of (theta, phi) pixel centres in RING order
    (cora.util.hputil.ang_positions = healpy.pix2ang)."""
    z, nphi, phi0, start = ring_info(nside)
    npix = 12 * nside**2
    out = np.empty((npix, 2))
    for r in range(z.size):
        sl = slice(start[r], start[r] + nphi[r])
        out[sl, 0] = np.arccos(z[r])
        out[sl, 1] = phi0[r] + 2.0 * np.pi * np.arange(nphi[r]) / nphi[r]
    return out


# ----------------------------------------------------------------------------
# coordinate helpers (cora.util.coord, restated)
# ----------------------------------------------------------------------------
def sph_to_cart(sph):
    sph = np.asarray(sph)
    st = np.sin(sph[..., 0])
    return np.stack([st * np.cos(sph[..., 1]), st * np.sin(sph[..., 1]), np.cos(sph[..., 0])], axis=-1)


def thetaphi_plane_cart(sph):
    sph = np.asarray(sph)
    t, p = sph[..., 0], sph[..., 1]
    that = np.stack([np.cos(t) * np.cos(p), np.cos(t) * np.sin(p), -np.sin(t)], axis=-1)
    phat = np.stack([-np.sin(p), np.cos(p), np.zeros_like(p)], axis=-1)
    return that, phat


# ----------------------------------------------------------------------------
# pixel kernels (_fast_tools.pyx)
# ----------------------------------------------------------------------------
def fringe(angpos, zenith, uv):
    """exp(2 pi i n.(u uhat + v vhat)), uhat = phihat(zenith), vhat = -thetahat(zenith)
    (_fast_tools.pyx:18-82)."""
    that, phat = thetaphi_plane_cart(zenith)
    uv3 = uv[0] * phat + uv[1] * (-that)
    phase = 2.0 * np.pi * (sph_to_cart(angpos) @ uv3)
    return np.cos(phase) + 1j * np.sin(phase)


def horizon(angpos, zenith):
    """signbit(-n.zenith): True above the horizon (visibility.py:27-46)."""
    proj = np.sum(sph_to_cart(angpos) * sph_to_cart(zenith), axis=-1)
    return np.signbit(-proj)


def beam_exptan(sintheta, fwhm):
    """exp(-alpha s^2 / (1 - s^2 + 1e-100)), alpha = ln2 / (2 tan^2(fwhm/2))
    (_fast_tools.pyx:248-282)."""
    alpha = np.log(2.0) / (2.0 * np.tan(fwhm / 2.0) ** 2)
    s2 = sintheta**2
    return np.exp(-alpha * s2 / (1.0 - s2 + 1e-100))


def construct_pol_real(beami, beamj, fr, hz):
    """Four Stokes response maps for real field patterns (_fast_tools.pyx:96-164)."""
    n = beami.shape[0]
    om_i = np.sum(hz * (beami[:, 0] ** 2 + beami[:, 1] ** 2)) * 4.0 * np.pi / n
    om_j = np.sum(hz * (beamj[:, 0] ** 2 + beamj[:, 1] ** 2)) * 4.0 * np.pi / n
    tc = fr * hz / np.sqrt(om_i * om_j)
    out = np.empty((4, n), dtype=np.complex128)
    out[0] = tc * (beami[:, 0] * beamj[:, 0] + beami[:, 1] * beamj[:, 1])
    out[1] = tc * (beami[:, 0] * beamj[:, 0] - beami[:, 1] * beamj[:, 1])
    out[2] = tc * (beami[:, 0] * beamj[:, 1] + beami[:, 1] * beamj[:, 0])
    out[3] = 1j * tc * (beami[:, 0] * beamj[:, 1] - beami[:, 1] * beamj[:, 0])
    return out


def construct_pol_complex(beami, beamj, fr, hz):
    """Four Stokes response maps for complex field patterns: the second beam enters conjugated, the solid angles are
    sums of |b|^2 (_fast_tools.pyx:167-242)."""
    n = beami.shape[0]
    om_i = np.sum(hz * (np.abs(beami[:, 0]) ** 2 + np.abs(beami[:, 1]) ** 2)) * 4.0 * np.pi / n
    om_j = np.sum(hz * (np.abs(beamj[:, 0]) ** 2 + np.abs(beamj[:, 1]) ** 2)) * 4.0 * np.pi / n
    tc = fr * hz / np.sqrt(om_i * om_j)
    cj = beamj.conj()
    out = np.empty((4, n), dtype=np.complex128)
    out[0] = tc * (beami[:, 0] * cj[:, 0] + beami[:, 1] * cj[:, 1])
    out[1] = tc * (beami[:, 0] * cj[:, 0] - beami[:, 1] * cj[:, 1])
    out[2] = tc * (beami[:, 0] * cj[:, 1] + beami[:, 1] * cj[:, 0])
    out[3] = 1j * tc * (beami[:, 0] * cj[:, 1] - beami[:, 1] * cj[:, 0])
    return out


# ----------------------------------------------------------------------------
# cylinder beam model (cylbeam.py)
# ----------------------------------------------------------------------------
def natural_spline(x, y):
    """Second derivatives of the natural cubic spline through (x, y) — the published
    algorithm behind cora.util.cubicspline.Interpolater."""
    n = x.size
    y2 = np.zeros(n)
    u = np.zeros(n)
    for i in range(1, n - 1):
        sig = (x[i] - x[i - 1]) / (x[i + 1] - x[i - 1])
        p = sig * y2[i - 1] + 2.0
        y2[i] = (sig - 1.0) / p
        u[i] = (y[i + 1] - y[i]) / (x[i + 1] - x[i]) - (y[i] - y[i - 1]) / (x[i] - x[i - 1])
        u[i] = (6.0 * u[i] / (x[i + 1] - x[i - 1]) - sig * u[i - 1]) / p
    for k in range(n - 2, -1, -1):
        y2[k] = y2[k] * y2[k + 1] + u[k]
    return y2


def spline_eval(x, y, y2, xv):
    khi = np.clip(np.searchsorted(x, xv, side="left"), 1, x.size - 1)
    klo = khi - 1
    h = x[khi] - x[klo]
    a = (x[khi] - xv) / h
    b = (xv - x[klo]) / h
    return a * y[klo] + b * y[khi] + ((a**3 - a) * y2[klo] + (b**3 - b) * y2[khi]) * (h * h) / 6.0


def fraunhofer_cylinder(fwhm, width):
    """1-D Fraunhofer pattern of an exptan-illuminated cylinder of `width` wavelengths:
    spline knots (kx, fx, fx'') (cylbeam.py:52-95)."""
    res = 16
    num = 512
    hnum = 512 // 2 - 1
    ua = -1.0 * np.linspace(-1.0, 1.0, num, endpoint=False)[::-1]
    ax = beam_exptan(2 * ua / (1 + ua**2), fwhm)
    axe = np.zeros(res * num)
    axe[: (hnum + 2)] = ax[hnum:]
    axe[-hnum:] = ax[:hnum]
    fx = np.fft.fft(axe).real
    kx = 2 * np.fft.fftfreq(res * num, ua[1] - ua[0]) / width
    fx = np.fft.fftshift(fx) / fx.max()
    kx = np.fft.fftshift(kx)
    sel = np.abs(kx) < 1.1
    fx, kx = np.ascontiguousarray(fx[sel]), np.ascontiguousarray(kx[sel])
    return kx, fx, natural_spline(kx, fx)


def telescope_frame(zenith):
    """xhat (East), yhat (North), zhat (up) for an unrotated cylinder (cylbeam.py:129)."""
    that, phat = thetaphi_plane_cart(zenith)
    return phat, -that, sph_to_cart(zenith)


def beam_amp(angpos, zenith, width, fwhm_x, fwhm_y):
    """Cylinder amplitude pattern (cylbeam.py:101-147)."""
    xhat, yhat, zhat = telescope_frame(zenith)
    kx, fx, f2 = fraunhofer_cylinder(fwhm_x, width)
    cvec = sph_to_cart(angpos)
    hz = (cvec @ zhat > 0.0).astype(np.float64)
    return spline_eval(kx, fx, f2, cvec @ xhat) * beam_exptan(cvec @ yhat, fwhm_y) * hz


def polpattern(angpos, dipole):
    """Unit polarisation vectors of a dipole in the (thetahat, phihat) basis (cylbeam.py:10-42)."""
    that, phat = thetaphi_plane_cart(angpos)
    pv = np.stack([that @ dipole, phat @ dipole], axis=-1)
    nrm = np.hypot(pv[..., 0], pv[..., 1])
    nrm = np.where(nrm == 0.0, 1.0, nrm)
    return pv / nrm[..., None]


def beam_x(angpos, zenith, width, fwhm_e, fwhm_h):
    xhat, yhat, zhat = telescope_frame(zenith)
    return beam_amp(angpos, zenith, width, fwhm_e, fwhm_h)[:, None] * polpattern(angpos, xhat)


def beam_y(angpos, zenith, width, fwhm_e, fwhm_h):
    xhat, yhat, zhat = telescope_frame(zenith)
    return beam_amp(angpos, zenith, width, fwhm_h, fwhm_e)[:, None] * polpattern(angpos, yhat)


# ----------------------------------------------------------------------------
# spherical harmonic quadrature (restated HEALPix map2alm, iter = 0, equal weights)
# ----------------------------------------------------------------------------
def lambda_lm(lmax, m, z):
    """Normalised associated Legendre functions lambda_lm(z) = Y_lm(theta, 0) for
    l = m..lmax at every z; shape (lmax+1-m, z.size).  Standard three-term recurrence."""
    z = np.asarray(z, dtype=np.float64)
    st = np.sqrt((1.0 - z) * (1.0 + z))
    out = np.zeros((lmax + 1 - m, z.size))
    if m > lmax:
        return out
    # log of the sectoral prefactor to dodge underflow of sin^m for large m
    logpre = 0.5 * (np.log(2.0 * m + 1.0) - np.log(4.0 * np.pi))
    if m > 0:
        k = np.arange(1, m + 1)
        logpre += 0.5 * np.sum(np.log((2.0 * k - 1.0) / (2.0 * k)))
    with np.errstate(divide="ignore"):
        lmm = ((-1.0) ** m) * np.exp(logpre + m * np.log(st)) if m > 0 else np.full(z.size, np.exp(logpre))
    out[0] = lmm
    if lmax > m:
        out[1] = np.sqrt(2.0 * m + 3.0) * z * lmm
    for l in range(m + 2, lmax + 1):
        a = np.sqrt((4.0 * l * l - 1.0) / (l * l - m * m))
        b = np.sqrt(((l - 1.0) ** 2 - m * m) / (4.0 * (l - 1.0) ** 2 - 1.0))
        out[l - m] = a * (z * out[l - m - 1] - b * out[l - m - 2])
    return out


XSIGN = 1.0


def wx_lm(lmax, m, z):
    """Spin-2 ring functions W_lm, X_lm (HEALPix: W = -(2lam + -2lam)/2, X = -(2lam - -2lam)/2)
    for l = m..lmax; zero for l < 2.  Closed forms in lambda_lm, lambda_{l-1,m}
    (Kamionkowski, Kosowsky & Stebbins 1997, eq. 2.25 rewritten for normalised functions)."""
    lam = lambda_lm(lmax, m, z)
    z = np.asarray(z, dtype=np.float64)
    s2 = (1.0 - z) * (1.0 + z)
    W = np.zeros_like(lam)
    X = np.zeros_like(lam)
    for l in range(max(m, 2), lmax + 1):
        nl = 2.0 * np.sqrt(1.0 / ((l - 1.0) * l * (l + 1.0) * (l + 2.0)))
        lam_l = lam[l - m]
        lam_lm1 = lam[l - m - 1] if l - 1 >= m else np.zeros_like(z)
        c = np.sqrt((2.0 * l + 1.0) / (2.0 * l - 1.0) * (l * l - m * m))
        W[l - m] = -nl * (-((l - m * m) / s2 + 0.5 * l * (l - 1.0)) * lam_l + c * z / s2 * lam_lm1)
        X[l - m] = XSIGN * nl * (m / s2) * ((l - 1.0) * z * lam_l - c * lam_lm1)
    return W, X


def ring_dft(maps, nside, mlist, sign=+1):
    """G[m, ring, ...] = sum_j maps[..., pix(ring, j)] exp(sign i m phi_j) for every m in mlist."""
    z, nphi, phi0, start = ring_info(nside)
    maps = np.asarray(maps)
    lead = maps.shape[:-1]
    mlist = np.asarray(mlist)
    out = np.zeros((mlist.size, z.size) + lead, dtype=np.complex128)
    for r in range(z.size):
        phi = phi0[r] + 2.0 * np.pi * np.arange(nphi[r]) / nphi[r]
        tw = np.exp(sign * 1j * np.outer(mlist, phi))  # (nm, nphi)
        seg = maps[..., start[r] : start[r] + nphi[r]]
        out[:, r] = np.tensordot(tw, seg, axes=([1], [-1]))
    return out


def ring_dft_fft(maps, nside, mlist, sign=+1):
    """`ring_dft` with one FFT per ring instead of the explicit twiddle matrix (what a production SHT does; libsharp's
    ring transforms are FFTs): sum_j f_j exp(s i m (phi0 + 2 pi j / n)) = exp(s i m phi0) * n * ifft(f)[m mod n] for
    s = +1 (fft for s = -1).  Same result to rounding (tests/test_oracle_btgen.py); used by the CPU baseline of
    bench.py so that the host figure is not dominated by an O(npix * nm) loop the reference does not have."""
    z, nphi, phi0, start = ring_info(nside)
    maps = np.asarray(maps)
    lead = maps.shape[:-1]
    mlist = np.asarray(mlist)
    out = np.zeros((mlist.size, z.size) + lead, dtype=np.complex128)
    r = 0
    while r < z.size:   # runs of rings with the same number of pixels (the whole equatorial belt is one run)
        r1 = r
        while r1 + 1 < z.size and nphi[r1 + 1] == nphi[r] and start[r1 + 1] == start[r1] + nphi[r1]:
            r1 += 1
        n = int(nphi[r])
        seg = maps[..., start[r] : start[r1] + n].reshape(lead + (r1 - r + 1, n))
        spec = np.fft.ifft(seg, axis=-1) * n if sign > 0 else np.fft.fft(seg, axis=-1)     # (..., nring_run, n)
        pick = spec[..., np.mod(mlist, n)]                                                  # (..., nring_run, nm)
        ph = np.exp(sign * 1j * np.outer(phi0[r : r1 + 1], mlist))                          # (nring_run, nm)
        out[:, r : r1 + 1] = np.moveaxis(pick * ph, (-2, -1), (1, 0))
        r = r1 + 1
    return out


_TABLES = {}


def ring_tables(nside, lmax, am, polarised):
    """(lambda, W, X) of `lambda_lm` / `wx_lm` for l = am..lmax on every ring of `nside`, remembered at the largest lmax
    asked for (the recurrences run upwards in l: the tables of a smaller lmax are the leading rows).  Only the FFT forms
    below use the cache — several GB at nside 512; `clear_tables()` drops it."""
    key = (int(nside), int(am), bool(polarised))
    hit = _TABLES.get(key)
    if hit is None or hit[0] < lmax:
        z = ring_info(nside)[0]
        lam = lambda_lm(lmax, am, z)
        W, X = wx_lm(lmax, am, z) if polarised else (None, None)
        hit = (lmax, lam, W, X)
        _TABLES[key] = hit
    n = lmax + 1 - am
    return hit[1][:n], (hit[2][:n] if polarised else None), (hit[3][:n] if polarised else None)


def clear_tables():
    _TABLES.clear()


def ring_synth_fft(F, nside, ms):
    """maps[..., pix(ring, j)] = sum_m F[m, ring, ...] exp(-i m phi_j): the inverse of `ring_dft_fft(sign=+1)` on
    band-limited rings, one FFT per ring (m folded into the N bins of the ring first — on the polar rings with
    N <= 2 lmax several m share a bin, which is the aliasing a map-space refinement sees).  F: (nm, nring) + lead."""
    z, nphi, phi0, start = ring_info(nside)
    ms = np.asarray(ms)
    lead = F.shape[2:]
    npix = 12 * nside**2
    maps = np.zeros(lead + (npix,), dtype=np.complex128)
    r = 0
    while r < z.size:
        r1 = r
        while r1 + 1 < z.size and nphi[r1 + 1] == nphi[r]:
            r1 += 1
        n = int(nphi[r])
        ph = np.exp(-1j * np.outer(ms, phi0[r : r1 + 1]))                                   # (nm, nrun)
        Fr = F[:, r : r1 + 1] * ph.reshape(ph.shape + (1,) * len(lead))                      # (nm, nrun) + lead
        X = np.zeros((n, r1 - r + 1) + lead, dtype=np.complex128)
        np.add.at(X, np.mod(ms, n), Fr)                                                      # fold m into the bins
        seg = np.fft.fft(X, axis=0)                                                          # sum_q X[q] e^{-2 pi i q j / n}
        seg = np.moveaxis(seg, (0, 1), (-1, -2))                                             # lead + (nrun, n)
        maps[..., start[r] : start[r1] + n] = seg.reshape(lead + ((r1 - r + 1) * n,))
        r = r1 + 1
    return maps


def _rmat(tab, v):
    """real table @ complex array without numpy promoting the table to complex: the product of the real and imaginary
    parts in one real matmul.  tab (a, b) float64, v (b, k) complex128 -> (a, k) complex128."""
    v = np.ascontiguousarray(v)
    return (tab @ v.view(np.float64).reshape(v.shape[0], -1)).view(np.complex128)


def _analysis_fft(maps, nside, lmax, polarised, ring_w=None):
    """`_analysis` for every m in -lmax..lmax with one FFT per ring and cached tables: {m: (P, lmax + 1 - |m|)}."""
    z, nphi, phi0, start = ring_info(nside)
    w = 4.0 * np.pi / (12 * nside**2)
    wr = w * (np.ones(z.size) if ring_w is None else np.asarray(ring_w, dtype=np.float64))
    ms = np.arange(-lmax, lmax + 1)
    G = ring_dft_fft(maps, nside, ms, sign=+1)           # (nm, nring, P)
    out = {}
    for mi, m in enumerate(ms):
        am = abs(int(m))
        lam, W, X = ring_tables(nside, lmax, am, polarised)
        sgn = (-1.0) ** am if m < 0 else 1.0
        g = G[mi] * wr[:, None]                          # (nring, P)
        c = np.zeros((g.shape[1], lmax + 1 - am), dtype=np.complex128)
        if polarised:
            sx = -sgn if m < 0 else 1.0
            tv = _rmat(lam, g[:, [0, 3]])                # (L - am, 2): T and V
            wq = _rmat(W, g[:, [1, 2]])                  # W g_Q, W g_U
            xq = _rmat(X, g[:, [1, 2]])                  # X g_Q, X g_U
            c[0] = sgn * tv[:, 0]
            c[3] = sgn * tv[:, 1]
            c[1] = sgn * wq[:, 0] - 1j * sx * xq[:, 1]
            c[2] = sgn * wq[:, 1] + 1j * sx * xq[:, 0]
        else:
            c[0] = sgn * _rmat(lam, g[:, :1])[:, 0]
        out[int(m)] = c
    return out


def _synthesis_fft(coef, nside, lmax, polarised, npol):
    """`_synthesis` with cached tables and one FFT per ring."""
    nring = 4 * nside - 1
    ms = np.array(sorted(coef))
    F = np.zeros((ms.size, nring, npol), dtype=np.complex128)
    for mi, m in enumerate(ms):
        c = coef[int(m)]                                  # (P, L - am)
        am = abs(int(m))
        lam, W, X = ring_tables(nside, lmax, am, polarised)
        sgn = (-1.0) ** am if m < 0 else 1.0
        if polarised:
            sx = -sgn if m < 0 else 1.0
            tv = _rmat(lam.T, c[[0, 3]].T)               # (nring, 2)
            we = _rmat(W.T, c[[1, 2]].T)                 # E W, B W
            xe = _rmat(X.T, c[[1, 2]].T)                 # E X, B X
            F[mi, :, 0] = sgn * tv[:, 0]
            F[mi, :, 3] = sgn * tv[:, 1]
            F[mi, :, 1] = sgn * we[:, 0] - 1j * sx * xe[:, 1]
            F[mi, :, 2] = sgn * we[:, 1] + 1j * sx * xe[:, 0]
        else:
            F[mi, :, 0] = sgn * _rmat(lam.T, c[:1].T)[:, 0]
    return ring_synth_fft(F, nside, ms)                   # (npol, npix)


def _analysis(maps, nside, lmax, polarised, ms, ring_w=None):
    """c[p, l, m] = sum_pix w f_p Y_lm(pix) for the m in `ms` (the reference's conj(SHT(conj f))); returns
    {m: (P, lmax + 1 - |m|)} with (T, E, B, V) for polarised maps."""
    z, nphi, phi0, start = ring_info(nside)
    npix = 12 * nside**2
    w = 4.0 * np.pi / npix
    rw = np.ones(z.size) if ring_w is None else np.asarray(ring_w, dtype=np.float64)
    G = ring_dft(maps, nside, ms, sign=+1)  # (nm, nring, P): conj-trick turns e^{-im phi} into e^{+im phi}
    out = {}
    for mi, m in enumerate(ms):
        am = abs(int(m))
        lam = lambda_lm(lmax, am, z) * (w * rw)  # (L-am, nring)
        sgn = (-1.0) ** am if m < 0 else 1.0
        g = G[mi]  # (nring, P)
        c = np.zeros((g.shape[1], lmax + 1 - am), dtype=np.complex128)
        c[0] = sgn * (lam @ g[:, 0])
        if polarised:
            W, X = wx_lm(lmax, am, z)
            W = W * (w * rw)
            X = X * (w * rw)
            # lambda_{l,-m} = (-1)^m lambda_lm ; W_{l,-m} = (-1)^m W_lm ; X_{l,-m} = -(-1)^m X_lm
            sx = -sgn if m < 0 else 1.0
            gq, gu = g[:, 1], g[:, 2]
            c[1] = sgn * (W @ gq) - 1j * sx * (X @ gu)
            c[2] = sgn * (W @ gu) + 1j * sx * (X @ gq)
            c[3] = sgn * (lam @ g[:, 3])
        out[int(m)] = c
    return out


def _synthesis(coef, nside, lmax, polarised, npol):
    """f_p(pix) = sum_lm c_lm conj(Y_lm(pix)) — the inverse of `_analysis` on band-limited maps (for the
    spin-2 pair the Hermitian block [[W, -iX], [iX, W]] applied once more)."""
    z, nphi, phi0, start = ring_info(nside)
    npix = 12 * nside**2
    maps = np.zeros((npol, npix), dtype=np.complex128)
    for m, c in coef.items():
        am = abs(m)
        lam = lambda_lm(lmax, am, z)
        sgn = (-1.0) ** am if m < 0 else 1.0
        F = np.zeros((npol, z.size), dtype=np.complex128)
        F[0] = sgn * (c[0] @ lam)
        if polarised:
            W, X = wx_lm(lmax, am, z)
            sx = -sgn if m < 0 else 1.0
            F[1] = sgn * (c[1] @ W) - 1j * sx * (c[2] @ X)
            F[2] = sgn * (c[2] @ W) + 1j * sx * (c[1] @ X)
            F[3] = sgn * (c[3] @ lam)
        for r in range(z.size):
            phi = phi0[r] + 2.0 * np.pi * np.arange(nphi[r]) / nphi[r]
            maps[:, start[r] : start[r] + nphi[r]] += F[:, r : r + 1] * np.exp(-1j * m * phi)[None, :]
    return maps


def alias_limits(nside, lmax, polarised, eps=1e-13):
    """Per north-cap ring i = 1, 2, ... the largest m whose ring functions still reach `eps` there (max over l <= lmax of
    |lambda_lm|, |W_lm|, |X_lm|), for the rings where that m is at least 2 i — the rings on which two m of the band can
    share a pixel-frequency bin (N = 4 i <= 2 mlim) with weights above eps.  Returns (mlim per alias ring, mcut)."""
    z = ring_info(nside)[0]
    mlim = []
    for i in range(1, nside):
        if 4 * i > 2 * lmax:
            break
        zi = z[i - 1 : i]
        best = -1
        for m in range(2 * i, lmax + 1):
            peak = np.abs(lambda_lm(lmax, m, zi)).max()
            if polarised:
                W, X = wx_lm(lmax, m, zi)
                peak = max(peak, np.abs(W).max(), np.abs(X).max())
            if peak >= eps:
                best = m
            elif m > best + 3:
                break
        if best < 2 * i:
            if i > len(mlim) + 8:
                break
            continue
        mlim += [-1] * (i - 1 - len(mlim)) + [best]
    return mlim, (max(mlim) if mlim else -1)


def refine_harmonic(a0, nside, lmax, polarised, niter, ring_w=None, eps=None):
    """healpy's refinement a <- a + A(map - S a) = a_0 + a - (A o S) a WITHOUT the map — the identity the device path
    (dm_bt_columns_iter) is built on, restated for the CPU tests.  The ring DFT of a synthesised ring is
    G_m'[r] = N_r sum_{m = m' mod N_r} e^{i (m' - m) phi0_r} F_m[r]: per m the Gram matrix of the ring functions under the
    quadrature, plus the alias terms of the polar rings with N_r <= 2 lmax.  With `eps` the alias terms are restricted to
    the rings and m of `alias_limits` (what the device does); None keeps all of them (exact).
    a0: {m: (P, lmax + 1 - |m|)} from `_analysis`; returns the refined dictionary."""
    z, nphi, phi0, start = ring_info(nside)
    w = 4.0 * np.pi / (12 * nside**2)
    wr = w * (np.ones(z.size) if ring_w is None else np.asarray(ring_w, dtype=np.float64))
    P = 4 if polarised else 1
    nring = z.size
    cull = None
    if eps is not None:
        ml, _ = alias_limits(nside, lmax, polarised, eps)
        cull = np.full(nring, -1)
        cull[: len(ml)] = ml
        cull[nring - len(ml) :] = ml[::-1]

    def synth(c, m):      # F_m[r], (P, nring)
        am = abs(m)
        lam, W, X = ring_tables(nside, lmax, am, polarised)
        sgn = (-1.0) ** am if m < 0 else 1.0
        F = np.zeros((P, nring), dtype=np.complex128)
        F[0] = sgn * (c[0] @ lam)
        if polarised:
            sx = -sgn if m < 0 else 1.0
            F[1] = sgn * (c[1] @ W) - 1j * sx * (c[2] @ X)
            F[2] = sgn * (c[2] @ W) + 1j * sx * (c[1] @ X)
            F[3] = sgn * (c[3] @ lam)
        return F

    def ana(g, m):        # g (P, nring) -> (P, L - |m|)
        am = abs(m)
        lam, W, X = ring_tables(nside, lmax, am, polarised)
        sgn = (-1.0) ** am if m < 0 else 1.0
        g = g * wr
        c = np.zeros((P, lmax + 1 - am), dtype=np.complex128)
        c[0] = sgn * (lam @ g[0])
        if polarised:
            sx = -sgn if m < 0 else 1.0
            c[1] = sgn * (W @ g[1]) - 1j * sx * (X @ g[2])
            c[2] = sgn * (W @ g[2]) + 1j * sx * (X @ g[1])
            c[3] = sgn * (lam @ g[3])
        return c

    a = {m: c.copy() for m, c in a0.items()}
    for _ in range(int(niter)):
        F = {m: synth(a[m], m) for m in a}
        new = {}
        for mp in a:
            G = F[mp] * nphi                                      # the k = 0 term on every ring: the Gram matrix
            for r in np.nonzero(nphi <= 2 * lmax)[0]:             # rings that can alias
                N = int(nphi[r])
                lim = lmax if cull is None else min(int(cull[r]), lmax)
                if abs(mp) > lim:
                    continue
                for k in range(int(np.ceil((mp - lim) / N)), int(np.floor((mp + lim) / N)) + 1):
                    if k:
                        m = mp - k * N
                        G[:, r] += N * np.exp(1j * (mp - m) * phi0[r]) * F[m][:, r]
            new[mp] = a0[mp] + a[mp] - ana(G, mp)
        a = new
    return a


def transfer_single(maps, nside, lmax, lside, polarised, mabs=None, niter=0, ring_w=None, fft=False):
    """The reference's ``_transfer_single``: conj(SHT(conj(map))) zero-embedded into
    (P, lside+1, 2*lside+1) with non-centred m (negative m wrapped to the end).

    maps: (npix,) complex for unpolarised, (4, npix) [I, Q, U, V] for polarised.
    mabs: optional list of |m|: only the columns +m and -m of those are filled (the rest stay zero) —
    the full-size parity tests compare a few m of a 3.1 Mpixel map and cannot afford all 1025 columns.
    niter, ring_w: healpy.map2alm's `iter` (Jacobi refinement: alm += analysis(map - synthesis(alm)), default 3 in
    healpy) and per-ring factors on the equal-area weight (`use_weights`).  What cora passes is NOT verifiable here
    (PARITY UNPINNED for this boundary); the default restates the plain equal-weight quadrature.
    """
    if niter or ring_w is not None:
        if mabs is not None:
            raise ValueError("the refinement needs every m of the map")
        P = 4 if polarised else 1
        m2 = np.asarray(maps).reshape(P, 12 * nside**2)
        ms = np.arange(-lmax, lmax + 1)
        # healpy.map2alm: alm = A(map); iter times: alm += A(map - S(alm)).  `fft` takes the forms with one FFT per ring
        # and remembered tables (what libsharp does; equal to the explicit sums to rounding, tests/test_oracle_btgen.py)
        ana = (lambda mp: _analysis_fft(mp, nside, lmax, polarised, ring_w)) if fft else \
              (lambda mp: _analysis(mp, nside, lmax, polarised, ms, ring_w))
        syn = _synthesis_fft if fft else _synthesis
        coef = ana(m2)
        for _ in range(int(niter)):
            res = m2 - syn(coef, nside, lmax, polarised, P)
            d = ana(res)
            coef = {m: coef[m] + d[m] for m in coef}
        out = np.zeros((P, lside + 1, 2 * lside + 1), dtype=np.complex128)
        for m, c in coef.items():
            out[:, abs(m) : lmax + 1, m if m >= 0 else 2 * lside + 1 + m] = c
        return out
    z, nphi, phi0, start = ring_info(nside)
    npix = 12 * nside**2
    w = 4.0 * np.pi / npix
    P = 4 if polarised else 1
    maps = np.asarray(maps).reshape(P, npix)
    out = np.zeros((P, lside + 1, 2 * lside + 1), dtype=np.complex128)
    ms = np.arange(-lmax, lmax + 1)
    if mabs is not None:
        keep = sorted({int(a) for a in mabs if 0 <= int(a) <= lmax})
        ms = np.array(sorted({-a for a in keep} | set(keep)), dtype=np.int64)
    # (nm, nring, P): conj-trick turns e^{-im phi} into e^{+im phi}
    G = (ring_dft_fft if fft else ring_dft)(maps, nside, ms, sign=+1)
    for mi, m in enumerate(ms):
        am = abs(m)
        lam = lambda_lm(lmax, am, z) * w  # (L-am, nring)
        sgn = (-1.0) ** am if m < 0 else 1.0
        g = G[mi]  # (nring, P)
        col = m if m >= 0 else 2 * lside + 1 + m
        out[0, am : lmax + 1, col] = sgn * (lam @ g[:, 0])
        if polarised:
            W, X = wx_lm(lmax, am, z)
            W = W * w
            X = X * w
            # lambda_{l,-m} = (-1)^m lambda_lm ; W_{l,-m} = (-1)^m W_lm ; X_{l,-m} = -(-1)^m X_lm
            sx = -sgn if m < 0 else 1.0
            gq, gu = g[:, 1], g[:, 2]
            out[1, am : lmax + 1, col] = sgn * (W @ gq) - 1j * sx * (X @ gu)
            out[2, am : lmax + 1, col] = sgn * (W @ gu) + 1j * sx * (X @ gq)
            out[3, am : lmax + 1, col] = sgn * (lam @ g[:, 3])
    return out


def fold_to_m(tarray, mmax):
    """(nfb, P, L, 2L-1) non-centred a_lm -> (mmax+1) arrays of (nfb, 2, P, L - m) in the
    reference's beam_m convention (beamtransfer.py:620-624, :663)."""
    nfb, P, L, _ = tarray.shape
    out = []
    for m in range(mmax + 1):
        blk = np.zeros((nfb, 2, P, L - m), dtype=np.complex128)
        blk[:, 0] = tarray[:, :, m:, m]
        if m > 0:
            blk[:, 1] = (-1) ** m * tarray[:, :, m:, -m].conj()
        out.append(blk)
    return out


# ----------------------------------------------------------------------------
# whole-telescope beam transfer generation for cylinder arrays
# ----------------------------------------------------------------------------
def max_lm(baselines, wavelengths, uwidth, vwidth=0.0):
    """drift/core/telescope.py:99-122."""
    umax = (np.abs(baselines[:, 0]) + uwidth) / wavelengths
    vmax = (np.abs(baselines[:, 1]) + vwidth) / wavelengths
    mmax = np.ceil(2 * np.pi * umax).astype(np.int64)
    lmax = np.ceil((mmax**2 + (2 * np.pi * vmax) ** 2) ** 0.5).astype(np.int64)
    return lmax, mmax


def beam_transfer_m(tel, mlist=None):
    """Generate beam_m blocks for a cylinder telescope description ``tel`` (a dict):

      polarised (bool), zenith (2,), baselines (B,2), uniquepairs (B,2), beamclass (nfeed,),
      wavelengths (F,), cylinder_width, fwhm_e, fwhm_h, lmax, mmax, l_boost,
      included_freq, included_baseline, accuracy_boost
      and optionally u_width, v_width, beam_fn(feed, freq_index, angpos) for non-cylinder telescope classes

    Returns {m: (F, 2, B, P, L) complex128} with the reference's ``beam_m`` layout
    (zero for l < m and for skipped frequencies / baselines).
    """
    pol = tel["polarised"]
    P = 4 if pol else 1
    F, B = tel["wavelengths"].size, tel["baselines"].shape[0]
    lside, mmax = int(tel["lmax"]), int(tel["mmax"])
    L = lside + 1
    mlist = list(range(mmax + 1)) if mlist is None else list(mlist)
    out = {m: np.zeros((F, 2, B, P, L), dtype=np.complex128) for m in mlist}
    cache = {}
    geo = {}
    for f in tel["included_freq"]:
        for b in tel["included_baseline"]:
            lm, _ = max_lm(tel["baselines"][b : b + 1], tel["wavelengths"][f], tel.get("u_width", tel.get("cylinder_width")),
                           tel.get("v_width", 0.0))
            lmax_bf = int(np.ceil(tel.get("l_boost", 1.0) * lm[0]))
            nside = nside_for_lmax(lmax_bf, tel.get("accuracy_boost", 1) if not pol else 1)
            if nside not in geo:
                ap = ang_positions(nside)
                geo[nside] = (ap, horizon(ap, tel["zenith"]).astype(np.float64))
            ap, hz = geo[nside]
            fi, fj = tel["uniquepairs"][b]
            beams = []
            for feed in (fi, fj):
                key = (nside, f, int(tel["beamclass"][feed]))
                if key not in cache and "beam_fn" in tel:
                    # any other telescope class: beam_fn(feed, freq index, angpos) is the class's beam() restated
                    # by the caller (telescope.py:954-973 caches by beam class like this)
                    cache[key] = np.asarray(tel["beam_fn"](int(feed), int(f), ap))
                if key not in cache:
                    width = tel["cylinder_width"] / tel["wavelengths"][f]
                    if not pol:
                        cache[key] = beam_amp(ap, tel["zenith"], width, tel["fwhm_h"], tel["fwhm_h"])
                    elif tel["beamclass"][feed] % 2 == 0:
                        cache[key] = beam_x(ap, tel["zenith"], width, tel["fwhm_e"], tel["fwhm_h"])
                    else:
                        cache[key] = beam_y(ap, tel["zenith"], width, tel["fwhm_e"], tel["fwhm_h"])
                beams.append(cache[key])
            uv = tel["baselines"][b] / tel["wavelengths"][f]
            fr = fringe(ap, tel["zenith"], uv)
            if pol and (np.iscomplexobj(beams[0]) or np.iscomplexobj(beams[1])):   # telescope.py:1278-1281
                maps = construct_pol_complex(beams[0].astype(np.complex128), beams[1].astype(np.complex128), fr, hz)
            elif pol:
                maps = construct_pol_real(beams[0], beams[1], fr, hz)
            else:
                pxarea = 4 * np.pi / ap.shape[0]
                om_i = np.sum(np.abs(beams[0]) ** 2 * hz) * pxarea
                om_j = np.sum(np.abs(beams[1]) ** 2 * hz) * pxarea
                maps = hz * fr * beams[0] * beams[1].conjugate() / np.sqrt(om_i * om_j)
            if tel.get("sht_iter", 0) or tel.get("sht_ring_weights") is not None:
                rw = tel.get("sht_ring_weights")
                t = transfer_single(maps, nside, lmax_bf, lside, pol, niter=int(tel.get("sht_iter", 0)),
                                    ring_w=None if rw is None else rw.get(int(nside)), fft=bool(tel.get("sht_fft", False)))
            else:
                t = transfer_single(maps, nside, lmax_bf, lside, pol, mabs=mlist, fft=bool(tel.get("sht_fft", False)))
            for m in mlist:
                out[m][f, 0, b, :, m:] = t[:, m:, m]
                if m > 0:
                    out[m][f, 1, b, :, m:] = (-1) ** m * t[:, m:, -m].conj()
    return out
