"""GPU side of the other telescope classes: host plug-in beams (beam / beamx / beamy) uploaded per beam class,
visibility-response maps against the unmodified reference's ``_beam_map_single`` (tests/golden/telescopes.npz),
and whole beam-transfer generation against the numpy oracle fed with the same beams."""
import os

import numpy as np
import pytest

import telescope_cases as tc

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def gold(golden_dir):
    return np.load(os.path.join(golden_dir, "telescopes.npz"))


@pytest.fixture(scope="module")
def ctx():
    from driftscan_amd import device

    device.reset_context()
    return device.get_context()


NAMES = ["gmrt", "restricted_box", "restricted_pol_gauss", "restricted_extra", "random", "gradient", "extra",
         "perturbed", "dish_pol"]


@pytest.mark.parametrize("name", NAMES)
def test_beams_and_maps_vs_reference(gold, ctx, name):
    from driftscan_amd import btgen, healpix

    t = tc.build(gold, name)
    nside = int(gold["nside"])
    npix = healpix.npix(nside)
    pol = t.num_pol_sky > 1
    ncomp = 2 if pol else 1
    t._init_trans(nside)
    hz = t._horizon.astype(np.float64)
    # beams of every class, the way btgen.fill_beam_m obtains them
    cth, sth = healpix.ring_trig(nside)
    frame = btgen.telescope_frame(t.zenith)
    feeds = gold[name + "_beam_feeds"]
    classes = [int(t.beamclass[f]) for f in feeds]
    beams = ctx.empty((len(feeds), npix * ncomp), np.float64)
    for k, (feed, bc) in enumerate(zip(feeds, classes)):
        spec = t.beam_spec(bc, 1)
        if spec is None:
            beams[k].copy_(ctx.to_device(t._beam_host(int(feed), 1, nside).reshape(-1)))
        else:
            ctx.bt_beam_cyl(nside, cth, sth, frame, spec[0], spec[1], spec[2], beams[k])
    ctx.sync()
    hb = beams.cpu().numpy().reshape((len(feeds), npix) + ((2,) if pol else ()))
    ref = gold[name + "_beams"] * (hz[None, :, None] if pol else hz[None, :])
    # derivative beams of the perturbed cylinder are finite differences at 1 % (x ~50 on the rounding of the patterns)
    tol = 1e-9 if name == "perturbed" else 1e-12
    assert np.abs(hb - ref).max() <= tol * max(np.abs(ref).max(), 1.0)
    # Stokes / intensity response maps of a few baselines
    bsel = gold[name + "_map_bl"]
    index = {bc: k for k, bc in enumerate(classes)}
    pairs = t.uniquepairs[bsel]
    bi = np.array([index[int(t.beamclass[i])] for i in pairs[:, 0]], dtype=np.int32)
    bj = np.array([index[int(t.beamclass[j])] for j in pairs[:, 1]], dtype=np.int32)
    uv = t.baselines[bsel] / t.wavelengths[1]
    maps = ctx.empty((len(bsel), 4 if pol else 1, npix), np.complex128)
    ctx.bt_maps(nside, cth, sth, frame, pol, beams, uv, bi, bj, maps)
    ctx.sync()
    hm = maps.cpu().numpy()
    refm = gold[name + "_maps"].reshape(hm.shape)
    scale = np.abs(refm).max()
    # fringe phases 2 pi u.n with |u| up to ~300 wavelengths (GMRT): rounding of the phase argument dominates
    mtol = 1e-8 if name == "perturbed" else (1e-10 if name == "gmrt" else 1e-11)
    assert np.abs(hm - refm).max() <= mtol * scale, np.abs(hm - refm).max() / scale


def _oracle_desc(t, beam_fn=None):
    d = dict(polarised=t.num_pol_sky > 1, zenith=t.zenith, baselines=t.baselines, uniquepairs=t.uniquepairs,
             beamclass=t.beamclass, wavelengths=t.wavelengths, lmax=t.lmax, mmax=t.mmax, l_boost=t.l_boost,
             included_freq=t.included_freq, included_baseline=t.included_baseline, accuracy_boost=t.accuracy_boost, sht_iter=t.sht_iter, sht_fft=True,
             u_width=t.u_width, v_width=t.v_width)
    if beam_fn is not None:
        d["beam_fn"] = beam_fn
    return d


def _check_beam_m(t, ctx, desc, mlist, tol=1e-10):
    from driftscan_amd import btgen
    from oracle import btgen as ob

    bm = btgen.beam_m_all(t, ctx=ctx).cpu().numpy()
    ref = ob.beam_transfer_m(desc, mlist=mlist)
    scale = max(np.abs(ref[m]).max() for m in ref)
    assert scale > 0
    for m in ref:
        assert np.abs(bm[m] - ref[m]).max() < tol * scale, (m, np.abs(bm[m] - ref[m]).max() / scale)
    return bm


@pytest.mark.parametrize("pol", [False, True])
def test_dish_array_beam_m_vs_oracle(ctx, pol):
    """Dish array end to end: closed-form dish beams on the host, everything else on the device."""
    from driftscan_amd import disharray

    cfg = dict(gridu=2, gridv=2, dish_width=2.0, num_freq=2, freq_lower=None, freq_upper=None, freq_start=400.0,
               freq_end=440.0, freq_mode="edge", tsys=1.0)
    t = (disharray.PolarisedDishArray if pol else disharray.UnpolarisedDishArray).from_config(cfg)

    def beam_fn(feed, f, ap):
        amp = disharray.beam_circular(ap, t.zenith, t.dish_width / t.wavelengths[f])
        if not pol:
            return amp
        return amp[:, None] * (np.array([0.0, 1.0]) if t.polarisation[feed] == "X" else np.array([1.0, 0.0]))

    bm = _check_beam_m(t, ctx, _oracle_desc(t, beam_fn), mlist=[0, 1, 7, t.mmax])
    assert bm.shape == (t.mmax + 1, t.nfreq, 2, t.nbase, t.num_pol_sky, t.lmax + 1)


def test_restricted_cylinder_beam_m_vs_oracle(ctx):
    """A windowed cylinder: device pattern -> host window -> device transfer, against the oracle's cylinder beam
    times the same window."""
    from driftscan_amd import restrictedcylinder
    from oracle import btgen as ob

    cfg = dict(num_freq=2, freq_start=400.0, freq_end=450.0, freq_mode="edge", num_cylinders=2, cylinder_width=2.0,
               num_feeds=3, feed_spacing=0.4, tsys=1.0, beam_type="gaussian", beam_height=35.0)
    t = restrictedcylinder.RestrictedCylinder.from_config(cfg)

    def beam_fn(feed, f, ap):
        d = ap - t.zenith[None, :]
        d = np.abs(np.where((d[:, 1] < np.pi)[:, None], d, d - np.array([0.0, 2 * np.pi])[None, :]))
        win = restrictedcylinder.gaussian_fwhm(d[:, 0], np.radians(t.beam_height))
        return win * ob.beam_amp(ap, t.zenith, t.cylinder_width / t.wavelengths[f], t.fwhm_h, t.fwhm_h)

    _check_beam_m(t, ctx, _oracle_desc(t, beam_fn), mlist=[0, 2, t.mmax])
    # and the products downstream of the beams run unchanged on such a telescope
    from driftscan_amd import btgen

    bm = btgen.beam_m_all(t, ctx=ctx, m_range=(1, 3))
    assert bm.shape[0] == 3


def test_focalplane_beam_m(ctx):
    """Zero-length 'baselines': the transfer of beam k is the harmonic transform of |beam_k|^2 / Omega_k —
    m = 0, l = 0 carries sqrt(4 pi) x (mean of the map) = 1 / sqrt(4 pi) x (4 pi / Omega) x Omega / (4 pi)."""
    from driftscan_amd import btgen, focalplane

    t = focalplane.FocalPlaneArray.from_config(dict(num_freq=2, freq_start=400.0, freq_end=450.0, beam_num_u=2,
                                                    beam_num_v=1, beam_spacing_u=20.0, beam_size=20.0, beam_pivot=400.0,
                                                    auto_correlations=True, force_lmax=40, force_mmax=40, sht_iter=0))
    assert t.nbase == 2
    bm = btgen.beam_m_all(t, ctx=ctx).cpu().numpy()
    # integral of (beam^2 / Omega) dOmega = 1  ->  a_00 = 1 / sqrt(4 pi): exact for the plain quadrature (sht_iter = 0:
    # Omega is the same pixel sum), and within the quadrature error of the map once healpy's refinement runs
    a00 = bm[0, :, 0, :, 0, 0]
    assert np.abs(a00 - 1.0 / np.sqrt(4 * np.pi)).max() < 1e-12
    t.sht_iter = 3
    a00r = btgen.beam_m_all(t, ctx=ctx).cpu().numpy()[0, :, 0, :, 0, 0]
    assert 1e-7 < np.abs(a00r - 1.0 / np.sqrt(4 * np.pi)).max() < 1e-3
    t.sht_iter = 0
    assert np.abs(bm[0, :, 0, :, 0, 0].imag).max() < 1e-14
    # the two beams point 20 degrees apart in azimuth: same |a_lm|, phases differ by exp(-i m dphi)
    m = 3
    dphi = np.radians(t.beam_spacing_u)
    with np.errstate(invalid="ignore", divide="ignore"):
        r = bm[m, 0, 0, 1, 0, m:] / bm[m, 0, 0, 0, 0, m:]
    big = np.abs(bm[m, 0, 0, 0, 0, m:]) > 1e-3 * np.abs(bm[m, 0, 0, 0, 0, m:]).max()
    assert np.abs(np.abs(r[big]) - 1).max() < 2e-2


def test_complex_patterns_pixel_kernel(ctx):
    """dm_bt_maps_c against the oracle's construct_pol_complex (pinned on the reference's compiled Cython in
    tests/test_oracle_btgen.py) and the complex unpolarised product b_i conj(b_j)."""
    from driftscan_amd import btgen, healpix
    from oracle import btgen as ob

    nside = 16
    npix = healpix.npix(nside)
    zen = np.array([np.pi / 2 - np.radians(45.0), 0.0])
    cth, sth = healpix.ring_trig(nside)
    frame = btgen.telescope_frame(zen)
    ap = ob.ang_positions(nside)
    hz = ob.horizon(ap, zen).astype(np.float64)
    rng = np.random.default_rng(3)
    width, fe, fh = 5.0 / 0.7, 2.0 * np.pi / 3.0 * 0.7, 2.0 * np.pi / 3.0
    bx = ob.beam_x(ap, zen, width, fe, fh) * np.exp(0.3j + 0.2j * np.cos(ap[:, 1]))[:, None]
    by = ob.beam_y(ap, zen, width, fe, fh) * np.exp(-0.7j + 0.4j * np.sin(ap[:, 0]))[:, None]
    hb = np.stack([bx * hz[:, None], by * hz[:, None]])                      # (2, npix, 2)
    beams = ctx.to_device(hb.reshape(2, -1))
    uv = np.array([[3.1, -2.2], [0.0, 1.7], [-4.0, 0.3]])
    bi, bj = np.array([0, 1, 1]), np.array([1, 1, 0])
    maps = ctx.empty((3, 4, npix), np.complex128)
    ctx.bt_maps(nside, cth, sth, frame, True, beams, uv, bi, bj, maps)
    ctx.sync()
    hm = maps.cpu().numpy()
    for k in range(3):
        ref = ob.construct_pol_complex(hb[bi[k]], hb[bj[k]], ob.fringe(ap, zen, uv[k]), hz)
        assert np.abs(hm[k] - ref).max() <= 1e-11 * np.abs(ref).max()
    # unpolarised: telescope.py:1156-1176 with complex beams
    ub = ob.beam_amp(ap, zen, width, fh, fh) * hz
    ub2 = np.stack([ub * np.exp(0.5j * ap[:, 1]), ub * np.exp(-0.2j * ap[:, 0])])
    m1 = ctx.empty((1, 1, npix), np.complex128)
    ctx.bt_maps(nside, cth, sth, frame, False, ctx.to_device(ub2), uv[:1], np.array([0]), np.array([1]), m1)
    ctx.sync()
    px = 4 * np.pi / npix
    om = [np.sum(np.abs(b) ** 2 * hz) * px for b in ub2]
    ref = hz * ob.fringe(ap, zen, uv[0]) * ub2[0] * ub2[1].conj() / np.sqrt(om[0] * om[1])
    assert np.abs(m1.cpu().numpy()[0, 0] - ref).max() <= 1e-11 * np.abs(ref).max()


@pytest.mark.parametrize("pol", [False, True])
def test_complex_beam_telescope_vs_oracle(ctx, pol):
    """A dish array whose apertures carry a phase gradient (complex field patterns): beam() -> complex host maps ->
    the fused path (dm_bt_columns_c: _construct_pol_complex inside the ring-transform kernels, FFT belt and caps) and the
    two-call path (dm_bt_maps_c -> dm_bt_sht), both against the oracle fed with the same patterns."""
    from driftscan_amd import btgen, disharray

    base = disharray.PolarisedDishArray if pol else disharray.UnpolarisedDishArray

    class PhasedDishes(base):
        complex_beams = True

        def _amplitude(self, freq):
            amp = base._amplitude(self, freq)
            return amp * np.exp(0.8j * np.sin(self._angpos[:, 0]) * np.cos(self._angpos[:, 1]))

    cfg = dict(gridu=2, gridv=2, dish_width=2.0, num_freq=2, freq_lower=None, freq_upper=None, freq_start=400.0,
               freq_end=440.0, freq_mode="edge", tsys=1.0)
    t = PhasedDishes.from_config(cfg)

    def beam_fn(feed, f, ap):
        amp = disharray.beam_circular(ap, t.zenith, t.dish_width / t.wavelengths[f])
        amp = amp * np.exp(0.8j * np.sin(ap[:, 0]) * np.cos(ap[:, 1]))
        if not pol:
            return amp
        return amp[:, None] * (np.array([0.0, 1.0]) if t.polarisation[feed] == "X" else np.array([1.0, 0.0]))

    _check_beam_m(t, ctx, _oracle_desc(t, beam_fn), mlist=[0, 1, 7, t.mmax])
    fused = btgen.beam_m_all(t, ctx=ctx).cpu().numpy()
    import os

    os.environ["DRIFTMI_BT_MAPS"] = "1"
    try:
        two_call = btgen.beam_m_all(t, ctx=ctx).cpu().numpy()
    finally:
        os.environ.pop("DRIFTMI_BT_MAPS", None)
    assert np.abs(fused - two_call).max() <= 1e-12 * np.abs(two_call).max()
    # a partition of m gives the same bits as the whole range (FFT belt + caps, complex patterns)
    part = btgen.beam_m_all(t, ctx=ctx, m_range=(3, 9)).cpu().numpy()
    assert np.array_equal(part, fused[3:10])

    # the declaration is not needed: complex maps returned by beam() are recognised (nothing drops the imaginary part)
    class Undeclared(PhasedDishes):
        complex_beams = False

    assert np.array_equal(btgen.beam_m_all(Undeclared.from_config(cfg), ctx=ctx).cpu().numpy(), fused)
