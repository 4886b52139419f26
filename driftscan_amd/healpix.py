"""HEALPix RING geometry on the host (tiny; the per-pixel work is on the GPU).

Restated from Gorski et al. (2005); replaces ``cora.util.hputil.ang_positions`` /
``nside_for_lmax`` used at drift/core/telescope.py:949, :1179-1184, :1288.
"""
import numpy as np


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


def ring_trig(nside):
    """(cos theta, sin theta) per ring, through theta = arccos(z) as pix2ang does."""
    theta = np.arccos(ring_z(nside))
    return np.cos(theta), np.sin(theta)


def npix(nside):
    return 12 * nside * nside
