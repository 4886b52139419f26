"""healpy.map2alm's `iter` carried out in harmonic space (dm_bt_columns_iter) — the setting of the reference's SHT that
cannot be read in this image (drift/core/telescope.py:1179-1191, :1288-1312 reach healpy through cora): GPU == oracle for
iter in {0, 3}, blocks independent of the partition of m, the map-free form against the map-space form, and the bracket
|beam_m(iter = 0) - beam_m(iter = 3)| on the reference's own test telescope (tests/testparams.yaml, the m = 14 block that
tests/test_functional.py:175-186 pins at approx(rel=1e-4, abs=1e-8))."""
import os
import subprocess
import sys

import numpy as np
import pytest
import yaml

pytestmark = pytest.mark.gpu

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.fixture(scope="module")
def ctx():
    from driftscan_amd._lib import Context

    c = Context(0, workspace_bytes=4 << 30)
    yield c
    c.close()


def _testparams_tel(golden_dir, **kw):
    from driftscan_amd import cylinder

    conf = yaml.safe_load(open(os.path.join(golden_dir, "testparams.yaml")))
    cfg = dict(conf["telescope"])
    cfg.pop("type", None)
    cfg.update(kw)
    return cylinder.PolarisedCylinderTelescope.from_config(cfg)


def _desc(t, fsel=None, bsel=None, **kw):
    d = dict(polarised=t.num_pol_sky > 1, zenith=t.zenith, baselines=t.baselines, uniquepairs=t.uniquepairs,
             beamclass=t.beamclass, wavelengths=t.wavelengths, cylinder_width=t.cylinder_width, fwhm_e=t.fwhm_e,
             fwhm_h=t.fwhm_h, lmax=t.lmax, mmax=t.mmax, l_boost=t.l_boost,
             included_freq=t.included_freq if fsel is None else fsel,
             included_baseline=t.included_baseline if bsel is None else bsel, accuracy_boost=t.accuracy_boost)
    d.update(kw)
    return d


def test_alias_info_is_a_function_of_the_group(ctx, golden_dir):
    from driftscan_amd import healpix
    from driftscan_amd._lib import bt_alias_info

    for nside, lmax, pol in ((128, 96, True), (128, 128, False), (512, 512, True), (8, 14, False)):
        cth, sth = healpix.ring_trig(nside)
        nr, mc = bt_alias_info(nside, cth, sth, pol, lmax)
        print("nside %d lmax %d pol %s: %d alias rings per cap, mcut %d" % (nside, lmax, pol, nr, mc))
        assert 0 < nr <= lmax // 2 + 1 and 2 * nr <= mc + 2 <= lmax + 2
        assert bt_alias_info(nside, cth, sth, pol, lmax) == (nr, mc)
    # the coupled m stay far below the band limit at the BASELINE sizes: a rank above mcut refines its own m only
    cth, sth = healpix.ring_trig(512)
    assert bt_alias_info(512, cth, sth, True, 512)[1] < 128


@pytest.mark.parametrize("pol", [False, True])
def test_partition_of_m_with_refinement(ctx, pol):
    """Any partition of m gives the same bits with iter = 3 (ranges that reach into the aliased m are served from the
    closure 0 .. mcut, ranges above it on their own)."""
    from driftscan_amd import btgen, cylinder

    cfg = dict(num_freq=2, freq_start=400.0, freq_end=420.0, freq_mode="edge", num_cylinders=2, cylinder_width=5.0,
               num_feeds=3, feed_spacing=0.5, tsys=1.0, sht_iter=3)
    cls = cylinder.PolarisedCylinderTelescope if pol else cylinder.UnpolarisedCylinderTelescope
    tel = cls.from_config(cfg)
    full = btgen.beam_m_all(tel, ctx=ctx).cpu().numpy()
    M = tel.mmax + 1
    edges = [0, 1, M // 4, M // 2, (3 * M) // 4 + 1, M]
    for lo, hi in zip(edges[:-1], edges[1:]):
        part = btgen.beam_m_all(tel, ctx=ctx, m_range=(lo, hi - 1)).cpu().numpy()
        assert np.array_equal(part, full[lo:hi]), (pol, lo, hi)
    # small column chunks (several calls per nside group) give the same bits as one call
    small = btgen.beam_m_all(tel, ctx=ctx, max_bytes=48 << 20).cpu().numpy()
    assert np.array_equal(small, full)


def test_testparams_bracket_and_oracle(ctx, golden_dir):
    """The reference's test telescope: what `iter` moves, and GPU == oracle at iter = 3 through the map-free path."""
    from driftscan_amd import btgen
    from oracle import btgen as ob

    t0 = _testparams_tel(golden_dir, sht_iter=0)
    t3 = _testparams_tel(golden_dir, sht_iter=3)
    b0 = btgen.beam_m_all(t0, ctx=ctx).cpu().numpy()
    b3 = btgen.beam_m_all(t3, ctx=ctx).cpu().numpy()
    for m in (0, 14, 40):
        scale = np.abs(b3[m]).max()
        diff = np.abs(b0[m] - b3[m])
        outside = diff > 1e-8 + 1e-4 * np.abs(b3[m])          # the reference's approx(rel=1e-4, abs=1e-8)
        nz = np.abs(b3[m]) > 0
        print("testparams m = %d: max |beam_m(iter 0) - beam_m(iter 3)| = %.2e of the block scale; %.1f %% of the non-zero "
              "entries outside approx(rel=1e-4, abs=1e-8)" % (m, diff.max() / scale, 100.0 * outside[nz].mean()))
        assert diff.max() > 1e-6 * scale      # the setting matters: horizon-cut beams are not band-limited
    # six columns against the oracle's map-space refinement
    fsel, bsel = np.array([0, t3.nfreq - 1]), np.array([0, t3.nbase // 2, t3.nbase - 1])
    ref = ob.beam_transfer_m(_desc(t3, fsel, bsel, sht_iter=3))
    scale = max(np.abs(ref[m]).max() for m in ref)
    worst = 0.0
    for m in range(t3.mmax + 1):
        worst = max(worst, np.abs(b3[m][fsel][:, :, bsel] - ref[m][fsel][:, :, bsel]).max() / scale)
    print("testparams iter = 3, six columns, all m: GPU (harmonic-space refinement) vs oracle (map-space) %.2e" % worst)
    assert worst < 1e-10
    # a range above the aliased m, and one inside them
    for lo, hi in ((70, 80), (3, 9)):
        part = btgen.beam_m_all(t3, ctx=ctx, m_range=(lo, hi)).cpu().numpy()
        assert np.array_equal(part, b3[lo:hi + 1])


@pytest.mark.parametrize("pol", [False, True])
def test_map_free_against_map_space_refinement(ctx, pol, tmp_path):
    """dm_bt_columns_iter (no maps) against the round-3 map-space refinement on the device (maps materialised, synthesis,
    inverse ring DFT, residual, re-analysis: DRIFTMI_BT_MAPS=1 DM_SHT_PIXEL_REFINE=1, read once per process)."""
    from driftscan_amd import btgen, cylinder

    cfg = dict(num_freq=2, freq_start=400.0, freq_end=450.0, freq_mode="edge", num_cylinders=2, cylinder_width=4.0,
               num_feeds=3, feed_spacing=0.5, tsys=1.0, sht_iter=3)
    cls = cylinder.PolarisedCylinderTelescope if pol else cylinder.UnpolarisedCylinderTelescope
    new = btgen.beam_m_all(cls.from_config(cfg), ctx=ctx).cpu().numpy()
    out = str(tmp_path / "old.npy")
    script = (
        "import sys, numpy as np\n"
        "sys.path.insert(0, %r)\n"
        "from driftscan_amd import btgen, cylinder\n"
        "cfg = %r\n"
        "cls = cylinder.PolarisedCylinderTelescope if %r else cylinder.UnpolarisedCylinderTelescope\n"
        "np.save(%r, btgen.beam_m_all(cls.from_config(cfg)).cpu().numpy())\n"
    ) % (ROOT, cfg, bool(pol), out)
    env = dict(os.environ, DRIFTMI_BT_MAPS="1", DM_SHT_PIXEL_REFINE="1")
    res = subprocess.run([sys.executable, "-c", script], env=env, capture_output=True, text=True, timeout=600)
    assert res.returncode == 0, res.stderr[-2000:]
    old = np.load(out)
    err = np.abs(new - old).max() / np.abs(old).max()
    print("pol %s: map-free vs map-space refinement %.2e" % (pol, err))
    assert err <= 1e-12
