"""GPU parity of BeamTransferNoSVD, BeamTransferFullSVD and BeamTransferTempSVD against
outputs of the unmodified reference classes (tests/golden/bt_variants.npz)."""
import os

import numpy as np
import pytest

from parity_util import assert_spectrum, relerr
from test_gpu_projections import FakeTelescope

pytestmark = pytest.mark.gpu


def _make(cls, g, tag, tmp, ms):
    from driftscan_amd import device, storage

    device.reset_context()
    F, B, P, lmax = (int(x) for x in g[tag + "_dims"])
    tel = FakeTelescope(F, B, P, lmax, g[tag + "_npower"])
    bt = cls(str(tmp), telescope=tel)
    bt._generate_dirs()
    for mi in ms:
        with storage.File(bt._mfile(mi), "w") as f:
            f.create_dataset("beam_m", data=g["%s_m%d_beam_m" % (tag, mi)][..., mi:])
    return tel, bt


def test_nosvd_covariances_and_kl(golden_dir, tmp_path):
    from driftscan_amd import beamtransfer, kltransform

    g = np.load(os.path.join(golden_dir, "bt_variants.npz"))
    ms = [int(m) for m in g["nosvd_mlist"]]
    tel, bt = _make(beamtransfer.BeamTransferNoSVD, g, "nosvd", tmp_path, ms)
    bt._generate_svdfiles()  # no-op by definition
    kl = kltransform.KLTransform.from_config(dict(threshold=0.1), bt, subdir="kl")
    kl._cvsg, kl._cvfg = g["nosvd_cv_sg"], g["nosvd_cv_fg"]
    for mi in ms:
        pre = "nosvd_m%d_" % mi
        assert bt.ndof(mi) == int(g[pre + "ndof"]) == bt.ndofmax
        cs, cn = kl.sn_covariance(mi)
        # telescope basis: no gauge freedom, the matrices themselves are comparable
        assert relerr(cs, g[pre + "cs"]) < 1e-12
        assert relerr(cn, g[pre + "cn"]) < 1e-12
        evals = kl._transform_m(mi)[0]
        # the un-compressed pencil is badly conditioned (no SVD cut): eigenvalues to 1e-6 of the largest,
        # the kept (S/N > threshold) ones relatively
        ref = g[pre + "evals"]
        assert_spectrum(evals, ref, 1e-6, "nosvd evals m=%d" % mi)
        v = g[pre + "vec_sky"]
        assert relerr(bt.project_vector_sky_to_svd(mi, v), g[pre + "sky_to_svd"]) < 1e-12
        back = bt.project_vector_svd_to_sky(mi, bt.project_vector_sky_to_svd(mi, v), conj=True)
        assert relerr(back, g[pre + "svd_to_sky_conj"]) < 1e-12
        with pytest.raises(NotImplementedError):
            bt.project_vector_svd_to_sky(mi, bt.project_vector_sky_to_svd(mi, v), temponly=True)


def test_fullsvd_products(golden_dir, tmp_path):
    from driftscan_amd import beamtransfer

    g = np.load(os.path.join(golden_dir, "bt_variants.npz"))
    F, B, P, lmax = (int(x) for x in g["fullsvd_dims"])
    ms = list(range(lmax + 1))
    tel, bt = _make(beamtransfer.BeamTransferFullSVD, g, "fullsvd", tmp_path, ms)
    assert bt.svd_len == min((lmax + 1) * P, 2 * B)
    bt._my_ms = lambda mlist_=None: ms
    bt._generate_svdfiles(regen=True)
    T = 2 * B
    for mi in ms:
        pre = "fullsvd_m%d_" % mi
        sv = bt.beam_singularvalues(mi)
        ref_sv = g[pre + "singularvalues"]
        assert sv.shape == ref_sv.shape
        assert np.abs(sv - ref_sv).max() < 1e-10 * ref_sv.max()
        bs, ref_bs = bt.beam_svd(mi), g[pre + "beam_svd"]
        but, ref_but = bt.beam_ut(mi), g[pre + "beam_ut"]
        for fi in range(F):
            n = int((ref_sv[fi] > 1e-10 * ref_sv[fi].max()).sum()) if ref_sv[fi].max() > 0 else 0
            if n == 0:
                continue
            X, Xr = bs[fi, :n].reshape(n, -1), ref_bs[fi, :n].reshape(n, -1)
            # gauge-free: the Gram matrix over the sky index and the projector of beam_ut
            assert relerr(X.conj().T @ X, Xr.conj().T @ Xr) < 1e-9
            U, Ur = but[fi, :n], ref_but[fi, :n]
            assert relerr(U.conj().T @ U, Ur.conj().T @ Ur) < 1e-9
            ib = bt.invbeam_svd(mi)[fi].reshape(-1, bs.shape[1])[:, :n]
            assert np.abs(X @ ib - np.eye(n)).max() < 1e-7


def test_tempsvd_products(golden_dir, tmp_path):
    """BeamTransferTempSVD (beamtransfer.py:1458-1593): SVD of the temperature block only, all polarisations projected."""
    from driftscan_amd import beamtransfer

    g = np.load(os.path.join(golden_dir, "bt_variants.npz"))
    F, B, P, lmax = (int(x) for x in g["tempsvd_dims"])
    ms = list(range(lmax + 1))
    tel, bt = _make(beamtransfer.BeamTransferTempSVD, g, "tempsvd", tmp_path, ms)
    assert bt.svd_len == min(lmax + 1, 2 * B)
    bt._my_ms = lambda mlist_=None: ms
    bt._generate_svdfiles(regen=True)
    L = lmax + 1
    for mi in ms:
        pre = "tempsvd_m%d_" % mi
        sv, ref_sv = bt.beam_singularvalues(mi), g[pre + "singularvalues"]
        assert sv.shape == ref_sv.shape
        assert np.abs(sv - ref_sv).max() < 1e-10 * max(ref_sv.max(), 1e-300)
        bs, ref_bs = bt.beam_svd(mi), g[pre + "beam_svd"]
        but, ref_but = bt.beam_ut(mi), g[pre + "beam_ut"]
        ib = bt.invbeam_svd(mi)
        assert bs.shape == ref_bs.shape and but.shape == ref_but.shape and ib.shape == g[pre + "invbeam_svd"].shape
        for fi in range(F):
            n = int((ref_sv[fi] > 1e-10 * ref_sv[fi].max()).sum()) if ref_sv[fi].max() > 0 else 0
            if n == 0:
                continue
            X, Xr = bs[fi, :n].reshape(n, -1), ref_bs[fi, :n].reshape(n, -1)
            assert relerr(X.conj().T @ X, Xr.conj().T @ Xr) < 1e-9           # gauge-free: rows up to phases
            U, Ur = but[fi, :n], ref_but[fi, :n]
            assert relerr(U.conj().T @ U, Ur.conj().T @ Ur) < 1e-9
            # the temperature block carries the singular values
            assert np.abs(np.linalg.norm(bs[fi, :n, 0, :], axis=1) - ref_sv[fi, :n]).max() < 1e-10 * ref_sv[fi].max()
            # pseudo-inverse of the whole projected beam: the same operator as the reference's la.pinv
            Pi, Pr = ib[fi].reshape(-1, bs.shape[1]) @ bs[fi].reshape(bs.shape[1], -1), \
                g[pre + "invbeam_svd"][fi].reshape(-1, bs.shape[1]) @ ref_bs[fi].reshape(bs.shape[1], -1)
            assert relerr(Pi, Pr) < 1e-7
