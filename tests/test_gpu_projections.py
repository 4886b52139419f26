"""GPU parity of the beam-transfer operators that do not depend on the SVD basis (SURVEY.md §8a
A17: invbeam_m, map-making, dirty back-projection, sky covariance -> visibility basis) against
outputs of the unmodified reference (tests/golden/projections.npz), plus the KL round trip
kl -> svd -> kl through the stored inverse modes."""
import os

import numpy as np
import pytest

from parity_util import relerr

pytestmark = pytest.mark.gpu


class FakeTelescope(object):
    def __init__(self, F, B, P, lmax, npower):
        self.nfreq, self.nbase, self.npairs = F, B, B
        self.num_pol_sky = P
        self.lmax = self.mmax = lmax
        self.included_freq = np.arange(F)
        self.included_baseline = np.arange(B)
        self.included_pol = np.arange(P)
        self.frequencies = np.linspace(400.0, 450.0, F)
        self.baselines = np.zeros((B, 2))
        self.tsys_flat = 1.0
        self._npower = npower

    def noisepower(self, bl_indices, f_indices, ndays=None):
        bl, fi = np.broadcast_arrays(bl_indices, f_indices)
        return self._npower[fi, bl]


@pytest.fixture(scope="module", params=["unpol", "pol"])
def setup(request, golden_dir, tmp_path_factory):
    from driftscan_amd import beamtransfer, device, storage

    device.reset_context()
    tag = request.param
    g = np.load(os.path.join(golden_dir, "projections.npz"))
    F, B, P, lmax = (int(x) for x in g[tag + "_dims"])
    tel = FakeTelescope(F, B, P, lmax, g[tag + "_npower"])
    bt = beamtransfer.BeamTransfer(str(tmp_path_factory.mktemp("proj_" + tag)), telescope=tel)
    bt._generate_dirs()
    mlist = [int(m) for m in g[tag + "_mlist"]]
    for mi in mlist:
        with storage.File(bt._mfile(mi), "w") as f:
            f.create_dataset("beam_m", data=g["%s_m%d_beam_m" % (tag, mi)][..., mi:])
    return g, bt, tag, mlist


def test_invbeam_and_mapmaking(setup):
    g, bt, tag, mlist = setup
    for mi in mlist:
        pre = "%s_m%d_" % (tag, mi)
        # pinv with a relative cut at 1e-6: the synthetic blocks have a clean gap there, so the
        # pseudo-inverse is well defined; 1e-9 relative allows for sigma_min^-1 ~ 1e4 amplification
        assert relerr(bt.invbeam_m(mi), g[pre + "invbeam_m"]) < 1e-9
        assert relerr(bt.project_vector_telescope_to_sky(mi, g[pre + "vec_tel"]), g[pre + "tel_to_sky"]) < 1e-9
        assert relerr(bt.project_vector_backward(mi, g[pre + "vec_tel"]), g[pre + "tel_to_sky"]) < 1e-9
        z = bt.project_vector_telescope_to_sky(mi, np.zeros_like(g[pre + "vec_tel"]))
        assert z.shape == g[pre + "tel_to_sky"].shape and not z.any()


def test_dirty_and_forward(setup):
    g, bt, tag, mlist = setup
    for mi in mlist:
        pre = "%s_m%d_" % (tag, mi)
        assert relerr(bt.project_vector_backward_dirty(mi, g[pre + "vec_tel"]), g[pre + "backward_dirty"]) < 1e-12
        assert relerr(bt.project_vector_sky_to_telescope(mi, g[pre + "vec_sky"]), g[pre + "sky_to_tel"]) < 1e-12


def test_matrix_sky_to_telescope(setup):
    g, bt, tag, mlist = setup
    cv = g[tag + "_cv"]
    for mi in mlist:
        pre = "%s_m%d_" % (tag, mi)
        assert relerr(bt.project_matrix_sky_to_telescope(mi, cv), g[pre + "mat_sky_to_tel"]) < 1e-12
        assert relerr(bt.project_matrix_sky_to_telescope(mi, cv, temponly=True), g[pre + "mat_sky_to_tel_temponly"]) < 1e-12


def test_kl_roundtrip_through_inverse_modes(golden_dir, tmp_path):
    """project_vector_kl_to_svd is the right inverse of project_vector_svd_to_kl on the kept modes."""
    from driftscan_amd import beamtransfer, device, kltransform, storage

    device.reset_context()
    g = np.load(os.path.join(golden_dir, "svdkl_unpol.npz"))
    import test_gpu_pipeline as tp

    tel = tp.FakeTelescope(g)
    bt = beamtransfer.BeamTransfer(str(tmp_path), telescope=tel)
    bt.polsvcut, bt.svcut = float(g["polsvcut"]), float(g["svcut"])
    bt._generate_dirs()
    mlist = [int(m) for m in g["mlist"]]
    for mi in mlist:
        with storage.File(bt._mfile(mi), "w") as f:
            f.create_dataset("beam_m", data=g["m%d_beam_m" % mi][..., mi:])
    bt._my_ms = lambda mlist_=None: mlist
    bt._generate_svdfiles(regen=True)
    kl = kltransform.KLTransform.from_config(dict(threshold=float(g["threshold"]), inverse=True), bt, subdir="klinv")
    kl._cvsg, kl._cvfg = g["cv_sg"], g["cv_fg"]
    kl._my_ms = lambda: mlist
    rng = np.random.default_rng(7)
    for mi in mlist:
        kl.transform_save(mi)
        evals, evecs = kl.modes_m(mi)
        if evals is None or evals.size == 0:
            continue
        v = rng.standard_normal(evals.size) + 1j * rng.standard_normal(evals.size)
        back = kl.project_vector_kl_to_svd(mi, v)
        assert back.shape == (evecs.shape[1],)
        again = kl.project_vector_svd_to_kl(mi, back)
        assert relerr(again, v) < 1e-8  # E inv(E) restricted to the kept rows; cond(E) enters
        with pytest.raises(Exception):
            kl.project_vector_kl_to_svd(mi, np.zeros(evals.size + 1, dtype=np.complex128))
