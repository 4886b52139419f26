"""Pin oracle/btgen.py: pixel kernels and beams against the reference's own outputs
(golden), the SHT against brute-force (spin-weighted) spherical-harmonic sums."""
import math
import os

import numpy as np
import pytest
import scipy.special as sp

from oracle import btgen as ob


@pytest.fixture(scope="module")
def pk(golden_dir):
    return np.load(os.path.join(golden_dir, "pixel_kernels.npz"))


def test_pixel_kernels_vs_reference(pk):
    ap, zen, uv = pk["angpos"], pk["zenith"], pk["uv"]
    assert np.abs(ob.fringe(ap, zen, uv) - pk["fringe"]).max() < 1e-12
    assert (ob.horizon(ap, zen) == pk["horizon"]).all()
    assert np.abs(ob.beam_exptan(pk["exptan_in"], float(pk["exptan_fwhm"])) - pk["exptan"]).max() < 1e-15
    hz = pk["horizon"].astype(np.float64)
    for key, bi, bj in (("pol_real_xy", "beam_x", "beam_y"), ("pol_real_xx", "beam_x", "beam_x")):
        out = ob.construct_pol_real(pk[bi], pk[bj], pk["fringe"], hz)
        assert np.abs(out - pk[key]).max() <= 1e-13 * np.abs(pk[key]).max()
    # complex field patterns: the reference's compiled _construct_pol_complex on phase-rotated beams
    outc = ob.construct_pol_complex(pk["beam_xc"], pk["beam_yc"], pk["fringe"], hz)
    assert np.abs(outc - pk["pol_complex_xy"]).max() <= 1e-13 * np.abs(pk["pol_complex_xy"]).max()
    # and it reduces to the real kernel on real patterns
    outr = ob.construct_pol_complex(pk["beam_x"].astype(complex), pk["beam_y"].astype(complex), pk["fringe"], hz)
    assert np.abs(outr - pk["pol_real_xy"]).max() <= 1e-13 * np.abs(pk["pol_real_xy"]).max()


def test_pixel_kernels_live_against_the_compiled_reference_extension():
    """oracle/_ref holds the reference's OWN native extension (drift/util/_fast_tools.pyx compiled by oracle/Makefile; a
    binary, git-ignored, it travels to the GPU box): the restated pixel kernels against it on seeds the fixture does not
    hold, at HEALPix pixel counts that exercise the OpenMP reduction."""
    from oracle import refimport

    ft = refimport.load_fast_tools()
    if ft is None:
        pytest.skip("oracle/_ref not built (make -C oracle ref needs the reference tree)")
    for seed, n in ((11, 1), (12, 257), (13, 12 * 32 * 32)):
        rng = np.random.default_rng(seed)
        ap = np.stack([np.arccos(rng.uniform(-1, 1, n)), rng.uniform(0, 2 * np.pi, n)], axis=-1)
        zen = np.array([np.pi / 2.0 - np.radians(rng.uniform(20.0, 60.0)), 0.0])
        uv = rng.uniform(-40.0, 40.0, 2)
        fr = ft.fringe(ap, zen, uv)
        assert np.abs(ob.fringe(ap, zen, uv) - fr).max() < 1e-11
        st = rng.uniform(-1, 1, n)
        assert np.abs(ob.beam_exptan(st, 0.9) - ft.beam_exptan(st, 0.9)).max() < 1e-15
        hz = ob.horizon(ap, zen).astype(np.float64)
        bx, by = rng.standard_normal((n, 2)), rng.standard_normal((n, 2))
        if hz.sum() == 0:
            continue
        ref = ft._construct_pol_real(bx, by, fr, hz)
        assert np.abs(ob.construct_pol_real(bx, by, fr, hz) - ref).max() <= 1e-12 * np.abs(ref).max()
        bxc, byc = bx * np.exp(0.3j), by * np.exp(-0.7j)
        refc = ft._construct_pol_complex(bxc, byc, fr, hz)
        assert np.abs(ob.construct_pol_complex(bxc, byc, fr, hz) - refc).max() <= 1e-12 * np.abs(refc).max()


def test_cylinder_beams_vs_reference(pk):
    ap, zen = pk["angpos"], pk["zenith"]
    w, fe, fh = float(pk["cyl_width"]), float(pk["cyl_fwhm_e"]), float(pk["cyl_fwhm_h"])
    kx, fx, f2 = ob.fraunhofer_cylinder(fh, w)
    assert np.abs(kx - pk["fraunhofer_x"]).max() < 1e-14
    assert np.abs(fx - pk["fraunhofer_y"]).max() < 1e-13
    assert np.abs(f2 - pk["fraunhofer_y2"]).max() < 1e-9 * np.abs(pk["fraunhofer_y2"]).max()
    assert np.abs(ob.beam_amp(ap, zen, w, fh, fh) - pk["beam_amp"]).max() < 1e-12
    assert np.abs(ob.beam_x(ap, zen, w, fe, fh) - pk["beam_x"]).max() < 1e-12
    assert np.abs(ob.beam_y(ap, zen, w, fe, fh) - pk["beam_y"]).max() < 1e-12


def test_ring_geometry():
    for nside in (1, 2, 4, 8):
        ap = ob.ang_positions(nside)
        assert ap.shape == (12 * nside**2, 2)
        # equal-area pixels: the z-moments of the pixel centres integrate low polynomials exactly
        z = np.cos(ap[:, 0])
        assert abs(z.mean()) < 1e-14
        assert abs((z**2).mean() - 1.0 / 3.0) < 0.05 / nside**2 + 1e-14
    # nside = 1: four pixels at z = 2/3 (phi = 45 deg + k 90), four on the equator (phi = k 90), four at z = -2/3
    ap = ob.ang_positions(1)
    assert np.allclose(np.cos(ap[:4, 0]), 2.0 / 3.0) and np.allclose(ap[:4, 1], np.pi / 4 + np.arange(4) * np.pi / 2)
    assert np.allclose(np.cos(ap[4:8, 0]), 0.0) and np.allclose(ap[4:8, 1], np.arange(4) * np.pi / 2)
    assert np.allclose(np.cos(ap[8:, 0]), -2.0 / 3.0)


def test_lambda_vs_scipy():
    z = np.linspace(-0.97, 0.97, 23)
    lmax = 40
    for m in (0, 1, 2, 7, 23, 40):
        lam = ob.lambda_lm(lmax, m, z)
        for l in (m, min(m + 1, lmax), min(m + 9, lmax), lmax):
            ref = sp.sph_harm_y(l, m, np.arccos(z), 0.0).real
            assert np.abs(lam[l - m] - ref).max() < 1e-12 * max(1.0, np.abs(ref).max())


def spin_ylm(s, l, m, theta, phi):
    """Goldberg et al. (1967) closed form of the spin-weighted spherical harmonics."""
    pref = (-1.0) ** m * math.sqrt(
        math.factorial(l + m) * math.factorial(l - m) * (2 * l + 1) / (4 * math.pi * math.factorial(l + s) * math.factorial(l - s))
    )
    out = np.zeros_like(theta, dtype=np.complex128)
    for r in range(0, l - s + 1):
        k = r + s - m
        if k < 0 or k > l + s:
            continue
        out += math.comb(l - s, r) * math.comb(l + s, k) * (-1.0) ** (l - r - s) * (1.0 / np.tan(theta / 2.0)) ** (2 * r + s - m)
    return pref * np.sin(theta / 2.0) ** (2 * l) * out * np.exp(1j * m * phi)


def test_spin2_functions_vs_goldberg():
    z = np.linspace(-0.9, 0.9, 11)
    th = np.arccos(z)
    lmax = 9
    for m in range(0, lmax + 1):
        W, X = ob.wx_lm(lmax, m, z)
        for l in range(max(m, 2), lmax + 1):
            f_p = spin_ylm(2, l, m, th, np.zeros_like(th)).real
            f_m = spin_ylm(-2, l, m, th, np.zeros_like(th)).real
            assert np.abs(W[l - m] + 0.5 * (f_p + f_m)).max() < 1e-11, (l, m)
            assert np.abs(X[l - m] + 0.5 * (f_p - f_m)).max() < 1e-11, (l, m)


def test_transfer_single_bruteforce_scalar():
    nside, lmax = 4, 7
    rng = np.random.default_rng(0)
    ap = ob.ang_positions(nside)
    npix = ap.shape[0]
    mp = rng.standard_normal(npix) + 1j * rng.standard_normal(npix)
    t = ob.transfer_single(mp, nside, lmax, lmax, False)[0]
    w = 4 * np.pi / npix
    for l in range(lmax + 1):
        for m in range(-l, l + 1):
            ylm = sp.sph_harm_y(l, m, ap[:, 0], ap[:, 1])
            # btrans = conj( sum w conj(map) conj(Y) ) = sum w map Y
            ref = w * np.sum(mp * ylm)
            assert abs(t[l, m] - ref) < 1e-12, (l, m)


def test_transfer_single_bruteforce_pol():
    nside, lmax = 4, 6
    rng = np.random.default_rng(1)
    ap = ob.ang_positions(nside)
    npix = ap.shape[0]
    maps = rng.standard_normal((4, npix)) + 1j * rng.standard_normal((4, npix))
    t = ob.transfer_single(maps, nside, lmax, lmax, True)
    w = 4 * np.pi / npix
    cm = maps.conj()  # the reference transforms the conjugated maps, real and imaginary parts separately
    for l in range(2, lmax + 1):
        for m in range(-l, l + 1):
            y2 = spin_ylm(2, l, m, ap[:, 0], ap[:, 1])
            ym2 = spin_ylm(-2, l, m, ap[:, 0], ap[:, 1])
            aE = aB = 0.0
            for part, fac in ((cm.real, 1.0), (cm.imag, 1.0j)):
                q, u = part[1], part[2]
                a2 = w * np.sum((q + 1j * u) * y2.conj())
                am2 = w * np.sum((q - 1j * u) * ym2.conj())
                aE = aE + fac * (-(a2 + am2) / 2.0)
                aB = aB + fac * (1j * (a2 - am2) / 2.0)
            assert abs(t[1, l, m] - np.conj(aE)) < 1e-11, ("E", l, m)
            assert abs(t[2, l, m] - np.conj(aB)) < 1e-11, ("B", l, m)
    # T and V are plain scalar transforms
    for p in (0, 3):
        for l in (0, 3, lmax):
            for m in (-l, 0, l):
                ref = w * np.sum(maps[p] * sp.sph_harm_y(l, m, ap[:, 0], ap[:, 1]))
                assert abs(t[p, l, m] - ref) < 1e-12


def test_sht_refinement_converges_on_band_limited_maps():
    """The oracle's synthesis is the inverse of its analysis on band-limited maps: healpy-style iterations drive
    the quadrature error of the equal-weight transform down (spin 0 and the spin-2 pair)."""
    from oracle import btgen as ob

    nside, lmax = 8, 10
    rng = np.random.default_rng(0)
    for pol in (False, True):
        P = 4 if pol else 1
        coef = {}
        for m in range(-lmax, lmax + 1):
            c = rng.standard_normal((P, lmax + 1 - abs(m))) + 1j * rng.standard_normal((P, lmax + 1 - abs(m)))
            if pol:
                c[1:3, : max(0, 2 - abs(m))] = 0.0      # no E/B below l = 2
            coef[m] = c
        maps = ob._synthesis(coef, nside, lmax, pol, P)
        errs = []
        for it in (0, 1, 3):
            t = ob.transfer_single(maps if pol else maps[0], nside, lmax, lmax, pol, niter=it, ring_w=np.ones(4 * nside - 1))
            errs.append(max(np.abs(t[:, abs(m):, m if m >= 0 else 2 * lmax + 1 + m] - coef[m]).max() for m in coef))
        assert errs[1] < 0.3 * errs[0] and errs[2] < 0.03 * errs[0], errs
        t0 = ob.transfer_single(maps if pol else maps[0], nside, lmax, lmax, pol)
        t1 = ob.transfer_single(maps if pol else maps[0], nside, lmax, lmax, pol, ring_w=np.ones(4 * nside - 1))
        assert np.array_equal(t0, t1)


def test_sht_against_healpy_fixture(golden_dir):
    """Consumes tests/golden/sht_healpy.npz (scratch/pin_sht_with_healpy.py, wherever healpy exists): the one
    route by which the SHT boundary can be pinned."""
    import os

    import pytest

    from oracle import btgen as ob

    path = os.path.join(golden_dir, "sht_healpy.npz")
    if not os.path.exists(path):
        pytest.skip("no healpy fixture: healpy / cora are not available in the build image (parity unpinned)")
    g = np.load(path)
    for nside in (16, 32):
        lmax = 3 * nside // 2
        m = g["map_n%d" % nside]
        for it in (0, 1, 3):
            key = "alm_n%d_iter%d_w0" % (nside, it)
            if key in g:
                t = ob.transfer_single(m, nside, lmax, lmax, False, niter=it, ring_w=np.ones(4 * nside - 1))[0]
                assert np.abs(t - g[key]).max() < 1e-10 * np.abs(g[key]).max(), key


def test_ring_dft_by_fft_matches_the_explicit_sum():
    """`ring_dft_fft` (one FFT per ring, used by the CPU baseline of bench.py) against the explicit twiddle sums of
    `ring_dft`, including |m| beyond the pixel count of the polar rings (aliasing) and negative m."""
    from oracle import btgen as ob

    rng = np.random.default_rng(11)
    nside = 8
    maps = rng.standard_normal((4, 12 * nside**2)) + 1j * rng.standard_normal((4, 12 * nside**2))
    ms = np.arange(-40, 41)
    for sign in (+1, -1):
        a = ob.ring_dft(maps, nside, ms, sign=sign)
        b = ob.ring_dft_fft(maps, nside, ms, sign=sign)
        assert a.shape == b.shape and np.abs(a - b).max() <= 1e-12 * np.abs(a).max()
    lmax = 12
    t0 = ob.transfer_single(maps, nside, lmax, lmax, True)
    t1 = ob.transfer_single(maps, nside, lmax, lmax, True, fft=True)
    assert np.abs(t0 - t1).max() <= 1e-12 * np.abs(t0).max()


def test_refinement_fft_forms_match_the_explicit_ones():
    """The map-space refinement with one FFT per ring and remembered tables (what the GPU tests at nside 128 - 512 can
    afford) against the explicit pixel sums, with polar rings that alias (N = 4 i <= 2 lmax) and ring weights."""
    from oracle import btgen as ob

    rng = np.random.default_rng(21)
    for nside, lmax in ((8, 11), (16, 23)):
        rw = 1.0 + 1e-3 * rng.standard_normal(4 * nside - 1)
        for pol in (False, True):
            P = 4 if pol else 1
            maps = rng.standard_normal((P, 12 * nside**2)) + 1j * rng.standard_normal((P, 12 * nside**2))
            mp = maps if pol else maps[0]
            for it, w in ((1, None), (3, rw)):
                a = ob.transfer_single(mp, nside, lmax, lmax + 2, pol, niter=it, ring_w=w)
                b = ob.transfer_single(mp, nside, lmax, lmax + 2, pol, niter=it, ring_w=w, fft=True)
                assert np.abs(a - b).max() <= 1e-12 * np.abs(a).max(), (nside, pol, it)
    ob.clear_tables()


def test_refinement_in_harmonic_space_is_the_map_space_one():
    """a <- a_0 + a - (A o S) a with A o S formed from the ring functions alone (Gram matrix per m + the alias terms of the
    polar rings) equals healpy's residual-map iteration: the identity behind dm_bt_columns_iter.  Restricting the alias
    terms to `alias_limits(eps)` changes the result by ~1e-3 eps."""
    from oracle import btgen as ob

    rng = np.random.default_rng(22)
    nside, lmax = 16, 23
    rw = 1.0 + 1e-3 * rng.standard_normal(4 * nside - 1)
    for pol in (False, True):
        P = 4 if pol else 1
        maps = rng.standard_normal((P, 12 * nside**2)) + 1j * rng.standard_normal((P, 12 * nside**2))
        ms = np.arange(-lmax, lmax + 1)
        for w in (None, rw):
            a0 = ob._analysis(maps, nside, lmax, pol, ms, w)
            ref = ob.transfer_single(maps if pol else maps[0], nside, lmax, lmax, pol, niter=3, ring_w=np.ones(4 * nside - 1) if w is None else w)
            scale = np.abs(ref).max()
            for eps, tol in ((None, 1e-13), (1e-13, 1e-13), (1e-6, 1e-7)):
                a = ob.refine_harmonic(a0, nside, lmax, pol, 3, ring_w=w, eps=eps)
                err = max(np.abs(ref[:, abs(m):, m if m >= 0 else 2 * lmax + 1 + m] - c).max() for m, c in a.items())
                assert err <= tol * scale, (pol, eps, err / scale)
    # without any alias term the iteration is visibly different: the polar rings matter
    none = ob.refine_harmonic(a0, nside, lmax, pol, 3, ring_w=rw, eps=1e300)
    err = max(np.abs(ref[:, abs(m):, m if m >= 0 else 2 * lmax + 1 + m] - c).max() for m, c in none.items())
    assert err > 1e-6 * scale
    ob.clear_tables()


def test_sht_iter_bracket_on_testparams(golden_dir):
    """What healpy's `iter` moves on the reference's own test telescope (tests/testparams.yaml): six (f, b) columns,
    iter = 0 (plain equal-weight quadrature) against iter = 3 (healpy's documented default, the default here), at the
    m-blocks 0, 14 (the block tests/test_functional.py:175-186 pins at approx(rel=1e-4, abs=1e-8)) and 40.  The beams are
    horizon-cut, not band-limited: the refinement is not a no-op, and the two settings are not interchangeable."""
    import os

    import yaml

    from driftscan_amd import cylinder
    from oracle import btgen as ob

    conf = yaml.safe_load(open(os.path.join(golden_dir, "testparams.yaml")))
    cfg = dict(conf["telescope"])
    cfg.pop("type", None)
    t = cylinder.PolarisedCylinderTelescope.from_config(cfg)
    assert t.sht_iter == 3                      # healpy's documented default
    fsel, bsel = np.array([0, t.nfreq - 1]), np.array([0, t.nbase // 2, t.nbase - 1])
    desc = dict(polarised=True, zenith=t.zenith, baselines=t.baselines, uniquepairs=t.uniquepairs,
                beamclass=t.beamclass, wavelengths=t.wavelengths, cylinder_width=t.cylinder_width, fwhm_e=t.fwhm_e,
                fwhm_h=t.fwhm_h, lmax=t.lmax, mmax=t.mmax, l_boost=t.l_boost, included_freq=fsel,
                included_baseline=bsel, accuracy_boost=t.accuracy_boost, sht_fft=True)
    ms = [0, 14, 40]
    b0 = ob.beam_transfer_m(dict(desc, sht_iter=0), mlist=ms)
    b3 = ob.beam_transfer_m(dict(desc, sht_iter=3), mlist=ms)
    moved = {}
    for m in ms:
        x0, x3 = b0[m][fsel][:, :, bsel], b3[m][fsel][:, :, bsel]
        scale = np.abs(x3).max()
        diff = np.abs(x0 - x3)
        nz = np.abs(x3) > 0
        outside = (diff > 1e-8 + 1e-4 * np.abs(x3))[nz].mean()
        moved[m] = diff.max() / scale
        print("testparams m = %d: max |beam_m(iter 0) - beam_m(iter 3)| = %.2e of the block scale, %.1f %% of the entries "
              "outside approx(rel=1e-4, abs=1e-8)" % (m, moved[m], 100 * outside))
        assert outside > 0.05
    assert moved[0] > 1e-3 and moved[14] > 1e-5 and moved[40] > 1e-5
    ob.clear_tables()


def test_alias_limits_of_the_library_match_the_oracle():
    """`dm_bt_alias_info` (host arithmetic of libdriftmi: which polar rings and which m the harmonic-space refinement couples)
    against the oracle's `alias_limits` at the same threshold — the two restate the same rule from the same recurrences."""
    from driftscan_amd import healpix
    from driftscan_amd._lib import bt_alias_info
    from oracle import btgen as ob

    for nside, lmax, pol in ((8, 14, False), (32, 47, True), (64, 96, True), (128, 96, False), (128, 191, True)):
        cth, sth = healpix.ring_trig(nside)
        nr, mc = bt_alias_info(nside, cth, sth, pol, lmax)
        ml, mcut = ob.alias_limits(nside, lmax, pol, eps=1e-13)
        assert nr == len(ml) and mc == min(mcut, lmax), (nside, lmax, pol, nr, len(ml), mc, mcut)
