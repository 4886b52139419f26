"""HEALPix RING geometry on the host (tiny; the per-pixel work is on the GPU).

Restated from Gorski et al. (2005); replaces ``cora.util.hputil.ang_positions`` /
``nside_for_lmax`` used at drift/core/telescope.py:949, :1179-1184, :1288.
"""
import numpy as np


import functools


@functools.lru_cache(maxsize=None)
def nside_for_lmax(lmax, accuracy_boost=1):
    return int(2 ** (accuracy_boost + np.ceil(np.log((lmax + 1) / 3.0) / np.log(2.0))))


def ring_z(nside):
    """cos(theta) of the 4*nside-1 iso-latitude rings, north to south."""
    i = np.arange(1, 4 * nside, dtype=np.float64)
    cap_n = 1.0 - i**2 / (3.0 * nside**2)
    belt = (2.0 * nside - i) * 2.0 / (3.0 * nside)
    j = 4.0 * nside - i
    cap_s = -(1.0 - j**2 / (3.0 * nside**2))
    return np.where(i < nside, cap_n, np.where(i <= 3 * nside, belt, cap_s))


@functools.lru_cache(maxsize=16)
def ring_trig(nside):
    """(cos theta, sin theta) per ring, through theta = arccos(z) as pix2ang does."""
    theta = np.arccos(ring_z(nside))
    return np.cos(theta), np.sin(theta)


def npix(nside):
    return 12 * nside * nside


def ang_positions(nside):
    """(npix, 2) pixel centres (theta, phi) in RING order (cora.util.hputil.ang_positions = healpy.pix2ang):
    the sky positions a telescope class's host-side ``beam(feed, freq)`` is evaluated on."""
    nphi, phi0, start = ring_layout(nside)
    theta = np.arccos(ring_z(nside))
    out = np.empty((npix(nside), 2))
    out[:, 0] = np.repeat(theta, nphi)
    j = np.arange(npix(nside)) - np.repeat(start, nphi)
    out[:, 1] = np.repeat(phi0, nphi) + 2.0 * np.pi * j / np.repeat(nphi, nphi)
    return out


# ---- sky maps <-> spherical harmonics for the consumers of the operators (timestream simulation, map-making) ----
# cora.util.hputil.sphtrans_sky / sphtrans_inv_sky (drift/pipeline/timestream.py:262, :295, :451, :717) are not
# available; these restate them on the equal-weight HEALPix quadrature of the rest of the package.  The forward
# transform runs on the device (dm_bt_sht: ring DFT + Legendre GEMMs); the inverse is a host loop over rings — it is
# called once per map, off the hot path.
def ring_layout(nside):
    """(nphi, phi0, start) per ring of the RING scheme."""
    i = np.arange(1, 4 * nside)
    north, belt = i < nside, (i >= nside) & (i <= 3 * nside)
    j = np.where(north, i, 4 * nside - i)
    nphi = np.where(belt, 4 * nside, 4 * j).astype(np.int64)
    phi0 = np.where(belt, np.where((i + nside) % 2 == 0, np.pi / (4.0 * nside), 0.0), np.pi / (4.0 * np.maximum(j, 1)))
    start = np.concatenate([[0], np.cumsum(nphi)[:-1]])
    return nphi, phi0, start


def lambda_lm(lmax, m, z):
    """Normalised associated Legendre functions lambda_lm(z), l = m..lmax: (lmax + 1 - m, len(z))."""
    z = np.asarray(z, dtype=np.float64)
    st = np.sqrt((1.0 - z) * (1.0 + z))
    out = np.zeros((lmax + 1 - m, z.size))
    if m > lmax:
        return out
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


def wx_lm(lmax, m, z):
    """Spin-2 ring functions W_lm, X_lm (zero for l < 2), HEALPix convention."""
    lam = lambda_lm(lmax, m, z)
    z = np.asarray(z, dtype=np.float64)
    s2 = (1.0 - z) * (1.0 + z)
    W, X = np.zeros_like(lam), np.zeros_like(lam)
    for l in range(max(m, 2), lmax + 1):
        nl = 2.0 * np.sqrt(1.0 / ((l - 1.0) * l * (l + 1.0) * (l + 2.0)))
        lam_l = lam[l - m]
        lam_lm1 = lam[l - m - 1] if l - 1 >= m else np.zeros_like(z)
        c = np.sqrt((2.0 * l + 1.0) / (2.0 * l - 1.0) * (l * l - m * m))
        W[l - m] = -nl * (-((l - m * m) / s2 + 0.5 * l * (l - 1.0)) * lam_l + c * z / s2 * lam_lm1)
        X[l - m] = nl * (m / s2) * ((l - 1.0) * z * lam_l - c * lam_lm1)
    return W, X


def sphtrans_sky(skymap, lmax):
    """Real sky maps [freq, pol, pixel] (pol = T or T,Q,U,V) -> a_lm [freq, pol(T,E,B,V), l, m >= 0]
    (cora.util.hputil.sphtrans_sky).  Runs on the GPU through dm_bt_sht."""
    from .device import get_context

    skymap = np.asarray(skymap, dtype=np.float64)
    nfreq, npol, npx = skymap.shape
    nside = int(round(np.sqrt(npx / 12.0)))
    if 12 * nside * nside != npx or npol not in (1, 4):
        raise ValueError("sphtrans_sky: need [freq, 1 or 4, 12 nside^2] maps")
    ctx = get_context()
    cth, sth = ring_trig(nside)
    maps = ctx.to_device(skymap.astype(np.complex128))
    L = lmax + 1
    out = ctx.zeros((L, nfreq, 2, 1, npol, L), np.complex128)
    ctx.bt_sht(nside, cth, sth, npol == 4, lmax, lmax, lmax, nfreq, 1, np.arange(nfreq), np.zeros(nfreq, dtype=np.int64),
               np.full(nfreq, lmax), maps, out)
    h = out.cpu().numpy()[:, :, 0, 0]              # (m, freq, pol, l): c_lm = sum w f Y_lm
    # a real map's standard a_lm = sum w f conj(Y_lm) is the conjugate of that
    return np.ascontiguousarray(h.conj().transpose(1, 2, 3, 0))


def sphtrans_inv_sky(alm, nside):
    """a_lm [freq, pol(T,E,B,V), l, m >= 0] -> real maps [freq, pol(T,Q,U,V), pixel] (cora.util.hputil.sphtrans_inv_sky)."""
    alm = np.asarray(alm, dtype=np.complex128)
    nfreq, npol, L, M = alm.shape
    lmax = L - 1
    z = ring_z(nside)
    nphi, phi0, start = ring_layout(nside)
    out = np.zeros((nfreq, npol, npix(nside)))
    for m in range(min(M, L)):
        lam = lambda_lm(lmax, m, z)                               # (L - m, nring)
        a = alm[:, :, m:, m]                                      # (freq, pol, L - m)
        F = np.zeros((nfreq, npol, z.size), dtype=np.complex128)
        F[:, 0] = a[:, 0] @ lam
        if npol == 4:
            W, X = wx_lm(lmax, m, z)
            # Q + iU pair from E, B: the adjoint of the analysis block [[W, -iX], [iX, W]] with Y instead of conj(Y)
            F[:, 1] = a[:, 1] @ W + 1j * (a[:, 2] @ X)
            F[:, 2] = a[:, 2] @ W - 1j * (a[:, 1] @ X)
            F[:, 3] = a[:, 3] @ lam
        fac = 1.0 if m == 0 else 2.0
        for r in range(z.size):
            phi = phi0[r] + 2.0 * np.pi * np.arange(nphi[r]) / nphi[r]
            out[:, :, start[r] : start[r] + nphi[r]] += fac * (F[:, :, r, None] * np.exp(1j * m * phi)[None, None, :]).real
    return out
