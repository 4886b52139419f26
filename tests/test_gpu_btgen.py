"""GPU parity of beam-transfer generation (through the C ABI) against the numpy oracle."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def ctx():
    from driftscan_amd._lib import Context

    c = Context(0, workspace_bytes=2 << 30)
    yield c
    c.close()


def _tel(pol, **kw):
    from driftscan_amd import cylinder

    cfg = dict(num_freq=2, freq_start=400.0, freq_end=450.0, freq_mode="edge", num_cylinders=2, cylinder_width=2.0,
               num_feeds=3, feed_spacing=0.4, tsys=1.0)
    cfg.update(kw)
    klass = cylinder.PolarisedCylinderTelescope if pol else cylinder.UnpolarisedCylinderTelescope
    return klass.from_config(cfg)


def _oracle_desc(t):
    return dict(polarised=t.num_pol_sky > 1, zenith=t.zenith, baselines=t.baselines, uniquepairs=t.uniquepairs,
                beamclass=t.beamclass, wavelengths=t.wavelengths, cylinder_width=t.cylinder_width, fwhm_e=t.fwhm_e,
                fwhm_h=t.fwhm_h, lmax=t.lmax, mmax=t.mmax, l_boost=t.l_boost, included_freq=t.included_freq,
                included_baseline=t.included_baseline, accuracy_boost=t.accuracy_boost, sht_iter=t.sht_iter, sht_fft=True)


def test_beam_and_maps_kernels(ctx):
    """Pixel kernels alone: beams and Stokes maps on a small nside."""
    from driftscan_amd import btgen, healpix
    from oracle import btgen as ob

    t = _tel(True)
    nside = 16
    cth, sth = healpix.ring_trig(nside)
    frame = btgen.telescope_frame(t.zenith)
    ap = ob.ang_positions(nside)
    npix = ap.shape[0]
    beams = ctx.empty((2, npix * 2), np.float64)
    for bc in (0, 1):
        kind, tab, fw = t.beam_spec(bc, 0)
        ctx.bt_beam_cyl(nside, cth, sth, frame, kind, tab, fw, beams[bc])
    ctx.sync()
    hb = beams.cpu().numpy().reshape(2, npix, 2)
    w = t.cylinder_width / t.wavelengths[0]
    rx = ob.beam_x(ap, t.zenith, w, t.fwhm_e, t.fwhm_h)
    ry = ob.beam_y(ap, t.zenith, w, t.fwhm_e, t.fwhm_h)
    assert np.abs(hb[0] - rx).max() < 1e-12 and np.abs(hb[1] - ry).max() < 1e-12
    uv = np.array([[3.1, -2.2], [0.0, 1.7]])
    maps = ctx.empty((2, 4, npix), np.complex128)
    ctx.bt_maps(nside, cth, sth, frame, True, beams, uv, np.array([0, 1]), np.array([1, 1]), maps)
    ctx.sync()
    hm = maps.cpu().numpy()
    hz = ob.horizon(ap, t.zenith).astype(np.float64)
    for k, (bi, bj) in enumerate(((rx, ry), (ry, ry))):
        ref = ob.construct_pol_real(bi, bj, ob.fringe(ap, t.zenith, uv[k]), hz)
        assert np.abs(hm[k] - ref).max() <= 1e-11 * np.abs(ref).max()


@pytest.mark.parametrize("pol", [False, True])
def test_beam_m_vs_oracle(ctx, pol):
    from driftscan_amd import btgen
    from oracle import btgen as ob

    t = _tel(pol)
    bm = btgen.beam_m_all(t, ctx=ctx).cpu().numpy()
    ref = ob.beam_transfer_m(_oracle_desc(t))
    scale = max(np.abs(ref[m]).max() for m in ref)
    worst = 0.0
    for m in range(t.mmax + 1):
        worst = max(worst, np.abs(bm[m] - ref[m]).max() / scale)
        assert np.abs(bm[m][..., :m]).max(initial=0.0) == 0.0  # compact-storage region is exactly zero
    assert worst < 1e-10, worst


def test_skips_and_transfer_matrices(ctx):
    from driftscan_amd import btgen
    from oracle import btgen as ob

    t = _tel(False, skip_freq=[0], skip_baselines=[1, 4])
    bm = btgen.beam_m_all(t, ctx=ctx).cpu().numpy()
    assert np.abs(bm[:, 0]).max() == 0.0 and np.abs(bm[:, :, :, [1, 4]]).max() == 0.0
    ref = ob.beam_transfer_m(_oracle_desc(t), mlist=[0, 3, t.mmax])
    scale = np.abs(ref[0]).max()
    for m in ref:
        assert np.abs(bm[m] - ref[m]).max() < 1e-10 * scale
    # the (l, m) "transfer_matrices" view of the same numbers
    bl, fi = np.array([0, 2]), np.array([1, 1])
    tm = t.transfer_matrices(bl, fi)
    full = ob.beam_transfer_m(_oracle_desc(_tel(False)), mlist=range(t.mmax + 1))
    for k in range(2):
        for m in (0, 2, t.mmax):
            assert np.abs(tm[k, 0, :, m] - full[m][fi[k], 0, bl[k], 0]).max() < 1e-10 * scale
            if m:
                assert np.abs(tm[k, 0, :, -m] - (-1) ** m * full[m][fi[k], 1, bl[k], 0].conj()).max() < 1e-10 * scale


def test_sht_m_range_matches_full():
    """dm_bt_sht_range over a partition of m reproduces the all-m result block by block: the same bits from every wide
    call, rounding between a narrow call (belt through the matrix form) and a wide one (belt through the FFT)."""
    from driftscan_amd import btgen, cylinder, device

    device.reset_context()
    ctx = device.get_context()
    for cfg in (dict(num_freq=2, freq_start=400.0, freq_end=420.0, freq_mode="edge", num_cylinders=2, cylinder_width=4.0,
                     num_feeds=3, feed_spacing=0.5, tsys=1.0),):
        for cls in (cylinder.UnpolarisedCylinderTelescope, cylinder.PolarisedCylinderTelescope):
            tel = cls.from_config(cfg)
            full = btgen.beam_m_all(tel, ctx=ctx).cpu().numpy()
            M = tel.mmax + 1
            edges = [0, 1, M // 3, (2 * M) // 3 + 1, M]
            for lo, hi in zip(edges[:-1], edges[1:]):
                if hi <= lo:
                    continue
                part = btgen.beam_m_all(tel, ctx=ctx, m_range=(lo, hi - 1)).cpu().numpy()
                assert part.shape[0] == hi - lo
                if hi - lo > 8 and M > 8:
                    assert np.array_equal(part, full[lo:hi]), (cls.__name__, lo, hi)
                else:
                    # a narrow call (<= DM_BT_NARROW = 8 m-values) takes the belt through the matrix form instead of the
                    # FFT: the same sums in another order
                    assert np.abs(part - full[lo:hi]).max() <= 2e-13 * np.abs(full).max(), (cls.__name__, lo, hi)
            # ... narrow calls of every width (they pick different instantiations of the ring-transform kernels)
            if M > 12:
                for lo, n in ((M // 2, 6), (M // 2 + 2, 2), (M // 3, 4), (M - 8, 8)):
                    part = btgen.beam_m_all(tel, ctx=ctx, m_range=(lo, lo + n - 1)).cpu().numpy()
                    assert np.abs(part - full[lo:lo + n]).max() <= 2e-13 * np.abs(full).max(), (cls.__name__, lo, n)


@pytest.mark.parametrize("pol", [False, True])
@pytest.mark.parametrize("niter", [0, 3])
def test_beam_m_sht_iterations_and_ring_weights(ctx, pol, niter):
    """healpy.map2alm's `iter` (Jacobi refinement, carried out in harmonic space on the device) and per-ring weight
    factors: GPU == oracle.  These are the two settings of the reference's SHT (through cora) that cannot be read in
    this image; the defaults are healpy's documented ones (iter = 3, equal weights)."""
    from driftscan_amd import btgen
    from oracle import btgen as ob

    t = _tel(pol, sht_iter=0)
    base = btgen.beam_m_all(t, ctx=ctx).cpu().numpy()
    rng = np.random.default_rng(5)
    weights = {}
    lmax_bf, _ = t.baseline_lmax(np.repeat(np.arange(t.nbase), t.nfreq), np.tile(np.arange(t.nfreq), t.nbase))
    for ns in set(int(x) for x in btgen._nside_of(t, lmax_bf)):
        weights[ns] = 1.0 + 1e-3 * rng.standard_normal(4 * ns - 1)
    for rw in (None, weights):
        if niter == 0 and rw is None:
            continue
        t.sht_iter, t.sht_ring_weights = niter, rw
        bm = btgen.beam_m_all(t, ctx=ctx).cpu().numpy()
        desc = _oracle_desc(t)
        desc["sht_iter"], desc["sht_ring_weights"] = niter, rw
        ref = ob.beam_transfer_m(desc)
        scale = max(np.abs(ref[m]).max() for m in ref)
        worst = max(np.abs(bm[m] - ref[m]).max() for m in range(t.mmax + 1)) / scale
        moved = max(np.abs(bm[m] - base[m]).max() for m in range(t.mmax + 1)) / scale
        print("pol %s iter %d ring weights %s: GPU vs oracle %.2e, moved from the default by %.2e" % (pol, niter, rw is not None, worst, moved))
        assert worst < 1e-10, worst
        assert moved > 1e-8          # the options do something
        if niter:
            # an m-range is served from the same refined coefficients
            part = btgen.beam_m_all(t, ctx=ctx, m_range=(1, 2)).cpu().numpy()
            assert np.array_equal(part, bm[1:3])


@pytest.mark.parametrize("pol", [False, True])
def test_ring_skip_is_below_rounding(ctx, pol, monkeypatch):
    """(m, ring) pairs whose Legendre table column is below 1e-18 are neither transformed nor stored (libsharp's `mlim`
    rule behind healpy.map2alm, here from the tables themselves: bt_ring_skip_lookup): the blocks with every pair
    computed (DM_BT_RING_SKIP=0, read per call) differ by less than an ulp of the largest coefficient, at every m."""
    from driftscan_amd import btgen

    t = _tel(pol, cylinder_width=4.0, num_feeds=3, feed_spacing=0.5)
    new = btgen.beam_m_all(t, ctx=ctx).cpu().numpy()
    monkeypatch.setenv("DM_BT_RING_SKIP", "0")
    old = btgen.beam_m_all(t, ctx=ctx).cpu().numpy()
    monkeypatch.delenv("DM_BT_RING_SKIP")
    assert old.shape == new.shape
    scale = np.abs(old).max()
    assert np.abs(new - old).max() <= 2e-16 * scale
    # ... and relative to each block's own scale (the high-m blocks are the ones that lose rings)
    for m in range(old.shape[0]):
        sm = np.abs(old[m]).max()
        if sm > 0:
            assert np.abs(new[m] - old[m]).max() <= 1e-14 * sm, m


@pytest.mark.parametrize("pol", [False, True])
def test_fft_belt_and_fold_against_the_matrix_form(ctx, pol, tmp_path):
    """The round-3 ring transform (FFT in LDS on the equatorial belt, matrix-form sums on the caps) and the north/south
    fold of the Legendre stage against the round-2 path (matrix form on every ring, no fold; DM_BT_FFT=0 DM_BT_FOLD=0 —
    the switches are read once per process, hence the child process), on the same telescope."""
    import os
    import subprocess
    import sys

    from driftscan_amd import btgen

    t = _tel(pol, cylinder_width=4.0, num_feeds=3, feed_spacing=0.5)
    new = btgen.beam_m_all(t, ctx=ctx).cpu().numpy()
    out = str(tmp_path / "old.npy")
    script = (
        "import sys, numpy as np\n"
        "sys.path.insert(0, %r)\n"
        "from driftscan_amd import btgen, cylinder\n"
        "cfg = dict(num_freq=2, freq_start=400.0, freq_end=450.0, freq_mode='edge', num_cylinders=2, cylinder_width=4.0,\n"
        "           num_feeds=3, feed_spacing=0.5, tsys=1.0)\n"
        "cls = cylinder.PolarisedCylinderTelescope if %r else cylinder.UnpolarisedCylinderTelescope\n"
        "np.save(%r, btgen.beam_m_all(cls.from_config(cfg)).cpu().numpy())\n"
    ) % (os.path.dirname(os.path.dirname(os.path.abspath(__file__))), bool(pol), out)
    env = dict(os.environ, DM_BT_FFT="0", DM_BT_FOLD="0")
    res = subprocess.run([sys.executable, "-c", script], env=env, capture_output=True, text=True, timeout=600)
    assert res.returncode == 0, res.stderr[-2000:]
    old = np.load(out)
    assert old.shape == new.shape
    assert np.abs(new - old).max() <= 2e-13 * np.abs(old).max()
