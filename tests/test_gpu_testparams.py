"""BASELINE configs[0]: the reference's own tests/testparams.yaml (PolarisedCylinder, 2 x 5 feeds, 8 channels,
polsvcut 1.0, KLTransform without foregrounds + DoubleKL + two power-spectrum estimators) through
ProductManager.from_config(...).generate(), checked stage by stage against the oracle on the same inputs.
The yaml is the reference's test data file, copied verbatim into tests/golden/."""
import os

import numpy as np
import pytest
import yaml

from parity_util import assert_spectrum, relerr

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def products(golden_dir, tmp_path_factory):
    from driftscan_amd import device, manager

    device.reset_context()
    conf = yaml.safe_load(open(os.path.join(golden_dir, "testparams.yaml")))
    out = tmp_path_factory.mktemp("testparams")
    conf["config"]["output_directory"] = str(out / "testdir")
    cfile = out / "params.yaml"
    cfile.write_text(yaml.dump(conf))
    pm = manager.ProductManager.from_config(str(cfile))
    pm.generate()
    return pm


def test_shapes_and_files(products):
    pm = products
    t, bt = pm.telescope, pm.beamtransfer
    assert t.num_pol_sky == 4 and t.nfreq == 8
    assert os.path.exists(bt.directory + "/beam_m/COMPLETED")
    assert set(pm.kltransforms) == {"kl", "dk"} and set(pm.psestimators) == {"ps1", "ps2"}
    assert pm.kltransforms["kl"].evals_all().shape == (t.mmax + 1, bt.ndofmax)
    for name, nb in (("ps1", 2), ("ps2", 4)):  # 3 k edges -> 2 bands; x 2 theta bands for ps2
        fisher, bias = pm.psestimators[name].fisher_bias()
        assert fisher.shape == (nb, nb) and bias.shape == (nb,)
        assert np.allclose(fisher, fisher.T)


def test_btgen_columns(products):
    from oracle import btgen as ob

    pm = products
    t, bt = pm.telescope, pm.beamtransfer
    fsel, bsel = np.array([0, t.nfreq - 1]), np.array([0, t.nbase // 2, t.nbase - 1])
    desc = dict(polarised=True, zenith=t.zenith, baselines=t.baselines, uniquepairs=t.uniquepairs,
                beamclass=t.beamclass, wavelengths=t.wavelengths, cylinder_width=t.cylinder_width, fwhm_e=t.fwhm_e,
                fwhm_h=t.fwhm_h, lmax=t.lmax, mmax=t.mmax, l_boost=t.l_boost, included_freq=fsel,
                included_baseline=bsel, accuracy_boost=t.accuracy_boost, sht_iter=t.sht_iter, sht_fft=True)
    ms = [0, 7, t.mmax // 2]
    ref = ob.beam_transfer_m(desc, mlist=ms)
    scale = max(np.abs(ref[0]).max(), 1e-300)
    for mi in ms:
        got = bt.beam_m(mi)[fsel][:, :, bsel]
        assert np.abs(got - ref[mi][fsel][:, :, bsel]).max() < 1e-9 * scale, mi


def test_svd_and_kl_against_oracle(products):
    from driftscan_amd import storage
    from oracle import kl as okl
    from oracle import svdchain as osvd

    pm = products
    t, bt = pm.telescope, pm.beamtransfer
    kl = pm.kltransforms["kl"]
    noisew = bt._noisew()[:, : t.nbase]
    npw = kl._npower(1.0)
    for mi in (0, 5, t.mmax // 2):
        ref = osvd.svd_m(bt.beam_m(mi), noisew, polsvcut=bt.polsvcut)   # the oracle chain on the SAME blocks
        sv = bt.beam_singularvalues(mi)
        assert_spectrum(sv, ref["singularvalues"], 1e-10, "sv m=%d" % mi)
        svnum, svb = bt._svd_num(mi)
        rnum, rb = osvd.svd_num(ref["singularvalues"], bt.svcut)
        assert np.array_equal(svnum, rnum) and np.array_equal(svb, rb)
        cs, cn = okl.sn_covariance(ref["beam_svd"], ref["beam_ut"], ref["singularvalues"], kl.signal(), kl.foreground(),
                                   npw, svcut=bt.svcut, use_foregrounds=False)
        ev, _, _ = okl.kl_transform_m(cs, cn)
        with storage.File(kl._evfile % mi, "r") as f:
            assert_spectrum(f["evals_full"][:], ev, 1e-9, "kl evals m=%d" % mi)
