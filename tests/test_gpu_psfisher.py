"""GPU parity of the exact Fisher matrix (PSExact) against the unmodified reference's
`_work_fisher_bias_m` (tests/golden/psfisher.npz, G8) and of the assembled `fisher.hdf5` against
the oracle summed over m."""
import os

import numpy as np
import pytest

import test_gpu_pipeline as tp
from parity_util import relerr

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def setup(golden_dir, tmp_path_factory):
    from driftscan_amd import beamtransfer, device, kltransform, psestimation, storage

    device.reset_context()
    g = np.load(os.path.join(golden_dir, "svdkl_unpol.npz"))
    p = np.load(os.path.join(golden_dir, "psfisher.npz"))
    tel = tp.FakeTelescope(g)
    bt = beamtransfer.BeamTransfer(str(tmp_path_factory.mktemp("psf")), telescope=tel)
    bt.polsvcut, bt.svcut = float(g["polsvcut"]), float(g["svcut"])
    bt._generate_dirs()
    mlist = [int(m) for m in g["mlist"]]
    for mi in mlist:
        with storage.File(bt._mfile(mi), "w") as f:
            f.create_dataset("beam_m", data=g["m%d_beam_m" % mi][..., mi:])
    bt._my_ms = lambda mlist_=None: mlist
    bt._generate_svdfiles(regen=True)
    kl = kltransform.KLTransform.from_config(dict(threshold=float(g["threshold"])), bt, subdir="kl")
    kl._cvsg, kl._cvfg = g["cv_sg"], g["cv_fg"]
    for mi in mlist:
        kl.transform_save(mi)
    ps = psestimation.PSExact.from_config(dict(threshold=float(p["ps_threshold"])), kl, subdir="ps")
    ps.clarray = p["clarray"]
    ps.k_center = np.arange(p["clarray"].shape[0], dtype=np.float64)
    return g, p, bt, kl, ps, mlist


def test_fisher_m_against_reference(setup):
    g, p, bt, kl, ps, mlist = setup
    batch = ps.fisher_bias_batch(mlist)
    for (fisher, bias), mi in zip(batch, mlist):
        ref = p["m%d_fisher" % mi]
        assert ps.num_evals(mi) == int(p["m%d_nmodes" % mi])
        # The Fisher matrix is basis independent (a trace over the kept KL subspace weighted by the
        # eigenvalues): the device modes differ from LAPACK's by phases / rotations inside degenerate
        # clusters only.  1e-8 relative: the modes themselves carry eps * cond(N) ~ 1e-11 .. 1e-9.
        assert relerr(fisher, ref) < 1e-8, mi
        assert not bias.any()
        one, _ = ps.fisher_bias_m(mi)
        assert relerr(one, fisher) < 1e-12  # batched == one at a time
        # Hermitian, non-negative diagonal
        assert np.abs(fisher - fisher.T.conj()).max() <= 1e-12 * np.abs(fisher).max()
        assert (fisher.diagonal().real >= 0).all()


def test_fisher_file(setup):
    from driftscan_amd import storage

    g, p, bt, kl, ps, mlist = setup
    ps.telescope.mmax = max(mlist)  # generate() walks 0..mmax: only the fixture's m have products

    class _KL(object):  # m without products contribute nothing, as `num_evals == 0` does in the reference
        pass

    orig_modes = kl.modes_m
    kl.modes_m = lambda mi, threshold=None, device=False: orig_modes(mi, threshold, device=device) if mi in mlist else (None, None)
    orig_ndof = bt.ndof
    bt.ndof = lambda mi: orig_ndof(mi) if mi in mlist else 0
    orig_dev = bt._dev_products
    orig_svn = bt._svd_num
    first = mlist[0]
    bt._dev_products = lambda mi: orig_dev(mi if mi in mlist else first)
    bt._svd_num = lambda mi: orig_svn(mi if mi in mlist else first)
    try:
        nb = int(p["clarray"].shape[0])
        ps.read_config(dict(bandtype="polar", num_theta=1,
                            k_bands=[dict(spacing="linear", start=0.0, stop=float(nb), num=nb + 1)]))
        clarray = p["clarray"]
        ps.clarray = clarray
        ps.generate(regen=True)
    finally:
        kl.modes_m, bt.ndof, bt._dev_products, bt._svd_num = orig_modes, orig_ndof, orig_dev, orig_svn
    fisher, bias = ps.fisher_bias()
    ref = sum(p["m%d_fisher" % mi] for mi in mlist).real
    assert relerr(fisher, ref) < 1e-8
    with storage.File(ps.psdir + "/fisher.hdf5", "r") as f:
        for k in ("fisher", "bias", "covariance", "errors", "correlation", "band_power", "k_start", "k_end",
                  "k_center", "theta_start", "theta_end", "theta_center", "k_bands", "theta_bands"):
            assert k in f, k
        assert f["fisher"].shape == (clarray.shape[0], clarray.shape[0])
