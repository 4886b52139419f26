"""Host logic (feed pairs, band limits, noise, frequencies) against the unmodified
reference's outputs (tests/golden/geometry.npz).  CPU only; bit-exact indexing."""
import ast
import os

import numpy as np
import pytest

from driftscan_amd import cylinder


@pytest.fixture(scope="module")
def geo(golden_dir):
    return np.load(os.path.join(golden_dir, "geometry.npz"))


@pytest.mark.parametrize("name", ["testparams", "cfg2", "cfg3", "cfg5", "skip", "noincyl"])
def test_feedpairs_and_friends(geo, name):
    cfg = {k: ast.literal_eval(v) for k, v in zip(geo[name + "_cfg_keys"], geo[name + "_cfg_vals"])}
    klass = cylinder.PolarisedCylinderTelescope if str(geo[name + "_kind"]) == "pol" else cylinder.UnpolarisedCylinderTelescope
    t = klass.from_config(cfg)
    assert np.array_equal(t.feedpositions, geo[name + "_feedpositions"])
    assert np.array_equal(t.beamclass, geo[name + "_beamclass"])
    # bit-exact indexing
    assert np.array_equal(t.uniquepairs, geo[name + "_uniquepairs"])
    assert np.array_equal(t.redundancy, geo[name + "_redundancy"])
    if name + "_feedmap" in geo:   # (left out of the fixture for the 512-feed telescope of configs[4]: 3 x 512 x 512)
        assert np.array_equal(t.feedmap, geo[name + "_feedmap"])
        assert np.array_equal(t.feedmask, geo[name + "_feedmask"])
        assert np.array_equal(t.feedconj, geo[name + "_feedconj"])
    assert np.array_equal(t.baselines, geo[name + "_baselines"])
    assert np.array_equal(t.frequencies, geo[name + "_frequencies"])
    assert np.allclose(t.wavelengths, geo[name + "_wavelengths"], rtol=1e-15)
    assert t.lmax == int(geo[name + "_lmax"]) and t.mmax == int(geo[name + "_mmax"])
    assert np.array_equal(t.included_freq, geo[name + "_included_freq"])
    assert np.array_equal(t.included_baseline, geo[name + "_included_baseline"])
    assert np.array_equal(t.included_pol, geo[name + "_included_pol"])
    assert np.allclose(t.zenith, geo[name + "_zenith"])
    bl = np.arange(t.nbase)
    npw = np.array([np.asarray(t.noisepower(bl, fi)).reshape(-1) for fi in range(t.nfreq)])
    assert np.allclose(npw, geo[name + "_noisepower"], rtol=1e-14, atol=0)
    bb, ff = [a.ravel() for a in np.meshgrid(bl, np.arange(t.nfreq), indexing="ij")]
    lm, mm = t.baseline_lmax(bb, ff)
    assert np.array_equal(lm.reshape(t.nbase, t.nfreq), geo[name + "_lmax_bf"])
    assert np.array_equal(mm.reshape(t.nbase, t.nfreq), geo[name + "_mmax_bf"])


def test_fraunhofer_table(golden_dir):
    pk = np.load(os.path.join(golden_dir, "pixel_kernels.npz"))
    kx, fx, f2 = cylinder.fraunhofer_table(float(pk["cyl_fwhm_h"]), float(pk["cyl_width"]))
    assert np.abs(kx - pk["fraunhofer_x"]).max() < 1e-14
    assert np.abs(fx - pk["fraunhofer_y"]).max() < 1e-13
    assert np.abs(f2 - pk["fraunhofer_y2"]).max() < 1e-9 * np.abs(pk["fraunhofer_y2"]).max()
