#!/usr/bin/env python3
"""Pin the spherical-harmonic transform against healpy — wherever healpy exists.

The reference reaches healpy.map2alm through cora.util.hputil.sphtrans_complex[_pol] (drift/core/telescope.py:
1179-1191, :1288-1312); neither package is part of the build image and the reference's golden tarball is
network-only, so SURVEY.md section 8(c) declares this boundary "parity unpinned".  This script produces the
fixture that would pin it:

    python scratch/pin_sht_with_healpy.py            # needs healpy (and cora, if present, for its own wrapper)

writes tests/golden/sht_healpy.npz with seeded complex maps (nside 16 and 32, unpolarised and I/Q/U/V) and, for
iter in {0, 1, 3} and use_weights in {False, True}, the coefficients conj(map2alm(conj map)) assembled exactly as
cora's sphtrans_complex does (real and imaginary parts transformed separately, a_{l,-m} = (-1)^m conj(a_lm) per
part).  If cora is importable its own `sphtrans_complex[_pol]` output is stored as well, together with the values
of its module-level `_weight` / `_iter` style settings when it has them — that is the missing fact.
tests/test_oracle_btgen.py::test_sht_against_healpy_fixture consumes the file when it is present and tells which
(iter, use_weights) pair reproduces cora.
"""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def full_alm(hp, re_map, lmax, **kw):
    """(lmax+1, 2 lmax+1) non-centred a_lm of a REAL map, negative m from the reality condition."""
    alm = hp.map2alm(np.ascontiguousarray(re_map), lmax=lmax, **kw)
    out = np.zeros((lmax + 1, 2 * lmax + 1), dtype=np.complex128)
    for m in range(lmax + 1):
        for l in range(m, lmax + 1):
            a = alm[hp.Alm.getidx(lmax, l, m)]
            out[l, m] = a
            if m > 0:
                out[l, -m] = (-1) ** m * np.conj(a)
    return out


def main():
    try:
        import healpy as hp
    except ImportError:
        print("healpy is not installed here: nothing written (this is the situation the fixture is meant to end)")
        return 1
    rng = np.random.default_rng(7)
    out = dict(healpy_version=hp.__version__)
    for nside in (16, 32):
        lmax = 3 * nside // 2
        npix = 12 * nside**2
        m = rng.standard_normal(npix) + 1j * rng.standard_normal(npix)
        out["map_n%d" % nside] = m
        for it in (0, 1, 3):
            for uw in (False, True):
                try:
                    c = m.conj()
                    a = full_alm(hp, c.real, lmax, iter=it, use_weights=uw) + 1j * full_alm(hp, c.imag, lmax, iter=it, use_weights=uw)
                    out["alm_n%d_iter%d_w%d" % (nside, it, int(uw))] = a.conj()
                except Exception as e:  # ring-weight files may be missing from a minimal healpy install
                    print("nside %d iter %d use_weights %s: %r" % (nside, it, uw, e))
    try:
        from cora.util import hputil

        for nside in (16, 32):
            lmax = 3 * nside // 2
            out["cora_n%d" % nside] = hputil.sphtrans_complex(out["map_n%d" % nside].conj(), centered=False, lmax=lmax, lside=lmax).conj()
        for name in ("_weight", "_iter"):
            if hasattr(hputil, name):
                out["cora" + name] = getattr(hputil, name)
    except ImportError:
        print("cora is not installed: only the plain healpy variants are stored")
    path = os.path.join(ROOT, "tests", "golden", "sht_healpy.npz")
    np.savez_compressed(path, **out)
    print("wrote", path)
    return 0


if __name__ == "__main__":
    sys.exit(main())
