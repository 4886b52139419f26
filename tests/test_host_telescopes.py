"""Host side of the other telescope classes (drift/telescope/{gmrt,restrictedcylinder,exotic_cylinder}.py,
examples/disharray/simplearray.py) against the unmodified reference (tests/golden/telescopes.npz): feed pairs
bit-exact, band limits, noise, and the closed-form beams on the HEALPix pixel centres.  CPU only."""
import os

import numpy as np
import pytest

from driftscan_amd import focalplane, healpix, disharray

import telescope_cases as tc

NAMES = ["gmrt", "restricted_box", "restricted_pol_gauss", "restricted_extra", "random", "gradient", "extra",
         "perturbed", "dish_pol"]


@pytest.fixture(scope="module")
def gold(golden_dir):
    return np.load(os.path.join(golden_dir, "telescopes.npz"))


def test_case_list(gold):
    assert tc.names(gold) == NAMES


@pytest.mark.parametrize("name", NAMES)
def test_geometry(gold, name):
    t = tc.build(gold, name)
    assert np.array_equal(t.feedpositions, gold[name + "_feedpositions"])
    assert np.array_equal(t.beamclass, gold[name + "_beamclass"])
    for attr in ("uniquepairs", "redundancy", "feedmap", "feedmask", "feedconj", "baselines"):
        assert np.array_equal(getattr(t, attr), gold[name + "_" + attr]), attr
    assert np.array_equal(t.frequencies, gold[name + "_frequencies"])
    assert t.lmax == int(gold[name + "_lmax"]) and t.mmax == int(gold[name + "_mmax"])
    assert t.num_pol_sky == int(gold[name + "_npol"])
    assert np.allclose(t.zenith, gold[name + "_zenith"], rtol=0, atol=1e-15)
    bl = np.arange(t.nbase)
    npw = np.array([np.asarray(t.noisepower(bl, fi)).reshape(-1) for fi in range(t.nfreq)])
    assert np.allclose(npw, gold[name + "_noisepower"], rtol=1e-14, atol=0)


@pytest.mark.parametrize("name", tc.HOST_ONLY)
def test_host_beams(gold, name):
    t = tc.build(gold, name)
    nside = int(gold["nside"])
    t._init_trans(nside)
    assert t._angpos.shape == (healpix.npix(nside), 2)
    for k, feed in enumerate(gold[name + "_beam_feeds"]):
        ref = gold[name + "_beams"][k]
        got = np.asarray(t.beam(int(feed), 1))
        assert got.shape == ref.shape
        assert np.abs(got - ref).max() <= 1e-14 * max(np.abs(ref).max(), 1.0)
        # what btgen uploads: the same pattern, zero below the horizon
        up = t._beam_host(int(feed), 1, nside)
        hz = t._horizon.astype(float)
        assert np.array_equal(up, got * (hz if got.ndim == 1 else hz[:, None]))


def test_horizon_matches_cylinder_frame(gold):
    """signbit(-n . zenith) of the host plug-in path and the device's n . zhat > 0 select the same pixels."""
    from driftscan_amd import btgen

    t = tc.build(gold, "dish_pol")
    t._init_trans(16)
    zhat = btgen.telescope_frame(t.zenith)[6:]
    n = np.stack([np.sin(t._angpos[:, 0]) * np.cos(t._angpos[:, 1]), np.sin(t._angpos[:, 0]) * np.sin(t._angpos[:, 1]),
                  np.cos(t._angpos[:, 0])], axis=1)
    assert np.array_equal(t._horizon, n @ zhat > 0.0)


def test_focalplane_array():
    """Unpinned (the reference class is abstract as shipped): one auto-correlation 'baseline' per beam."""
    t = focalplane.FocalPlaneArray.from_config(dict(num_freq=2, freq_start=400.0, freq_end=450.0, beam_num_u=3,
                                                    beam_num_v=2, beam_spacing_u=2.0, beam_spacing_v=3.0, beam_size=4.0,
                                                    auto_correlations=True))
    assert t.nfeed == 6 and t.nbase == 6
    assert np.array_equal(t.uniquepairs, np.stack([np.arange(6), np.arange(6)], axis=1))
    assert np.array_equal(t.redundancy, np.ones(6, dtype=int))
    assert not t.baselines.any()
    t._init_trans(32)
    b = t.beam(0, 0)
    pk = t._angpos[np.argmax(b)]
    from driftscan_amd.telescope import sph_dot
    assert np.arccos(min(sph_dot(pk, t.beam_pointings[0]), 1.0)) < np.radians(2.0)
    t.square_beam = True
    sq = t.beam(3, 1)
    assert set(np.unique(sq)) <= {0.0, 1.0} and sq.sum() > 0
    # without auto-correlations no pair survives the zero-length cut
    assert focalplane.FocalPlaneArray.from_config(dict(beam_num_u=2, beam_num_v=2)).nbase == 0


def test_unpolarised_dish_array():
    t = disharray.UnpolarisedDishArray.from_config(dict(gridu=3, gridv=2, dish_width=4.0, num_freq=2,
                                                        freq_lower=None, freq_upper=None, freq_start=400.0,
                                                        freq_end=420.0))
    assert t.nfeed == 6 and t.num_pol_sky == 1
    # a 3 x 2 grid: separations (du, dv) with du in 0..2, dv in -1..1, one orientation each, minus (0, 0)
    assert t.nbase == 7
    assert t.redundancy.sum() == 6 * 5 // 2
    t._init_trans(8)
    b = t._beam_host(0, 0, 8)
    assert b.shape == (768,) and b.max() <= 1.0 + 1e-12 and (b[~t._horizon] == 0).all()
