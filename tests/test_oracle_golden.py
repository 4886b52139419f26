"""Pin oracle/ (the numpy restatement) against golden vectors produced by the
unmodified reference (oracle/gen_golden.py).  CPU only."""
import os

import numpy as np
import pytest

from oracle import kl as okl
from oracle import svdchain as osvd

from parity_util import assert_same_rowspace, assert_spectrum, relerr


@pytest.fixture(scope="module")
def mops(golden_dir):
    return np.load(os.path.join(golden_dir, "matrix_ops.npz"))


@pytest.mark.parametrize("name", ["rand_wide", "rand_tall", "lowrank", "empty"])
@pytest.mark.parametrize("rtol", [1e-10, 1e-4, 0.0])
def test_matrix_image_nullspace(mops, name, rtol):
    A = mops["A_" + name]
    tag = "%s_r%g" % (name, rtol)
    img, s = osvd.matrix_image(A, rtol=rtol)
    nul, s2 = osvd.matrix_nullspace(A, rtol=rtol)
    assert img.shape == mops["img_" + tag].shape
    assert nul.shape == mops["nul_" + tag].shape
    assert_spectrum(s, mops["imgs_" + tag], 1e-12, "image spectrum")
    assert_spectrum(s2, mops["nuls_" + tag], 1e-12, "nullspace spectrum")
    if name == "lowrank" and rtol == 0.0:
        return  # rtol=0 keeps numerically-zero directions: basis is arbitrary
    if img.size:
        assert_same_rowspace(img.T.conj(), mops["img_" + tag].T.conj(), 1e-9, "image")
    if nul.size:
        assert_same_rowspace(nul.T.conj(), mops["nul_" + tag].T.conj(), 1e-9, "nullspace")


@pytest.fixture(scope="module", params=["unpol", "pol"])
def svdkl(request, golden_dir):
    return np.load(os.path.join(golden_dir, "svdkl_%s.npz" % request.param))


def _noisew(g):
    return g["npower"] ** -0.5


def test_svd_chain(svdkl):
    g = svdkl
    for mi in g["mlist"]:
        pre = "m%d_" % mi
        res = osvd.svd_m(g[pre + "beam_m"], _noisew(g), polsvcut=float(g["polsvcut"]))
        sv_ref = g[pre + "singularvalues"]
        assert_spectrum(res["singularvalues"], sv_ref, 1e-11, "singular values m=%d" % mi)
        svnum, svbounds = osvd.svd_num(res["singularvalues"], float(g["svcut"]))
        assert (svnum == g[pre + "svnum"]).all()
        assert (svbounds == g[pre + "svbounds"]).all()
        # gauge-invariant: B^H B over the kept modes, and U^H U
        for f in range(int(g["F"])):
            n = svnum[f]
            b0 = g[pre + "beam_svd"][f, :n].reshape(n, -1)
            b1 = res["beam_svd"][f, :n].reshape(n, -1)
            assert relerr(b1.T.conj() @ b1, b0.T.conj() @ b0) < 1e-9
            u0 = g[pre + "beam_ut"][f, :n]
            u1 = res["beam_ut"][f, :n]
            assert relerr(u1.T.conj() @ u1, u0.T.conj() @ u0) < 1e-9
            ib0 = g[pre + "invbeam_svd"][f].reshape(-1, g[pre + "invbeam_svd"].shape[-1])[:, :n]
            ib1 = res["invbeam_svd"][f].reshape(-1, res["invbeam_svd"].shape[-1])[:, :n]
            assert relerr(ib1 @ b1, ib0 @ b0) < 1e-7


def test_covariance_projection(svdkl):
    g = svdkl
    for mi in g["mlist"]:
        pre = "m%d_" % mi
        svnum, svbounds = g[pre + "svnum"], g[pre + "svbounds"]
        for key, cv, to in (("proj_sg", "cv_sg", False), ("proj_fg", "cv_fg", False), ("proj_sg_temponly", "cv_sg", True)):
            out = okl.project_matrix_sky_to_svd(g[pre + "beam_svd"], svnum, svbounds, g[cv], temponly=to)
            assert relerr(out, g[pre + key]) < 1e-13
        npw = np.concatenate([g["npower"], g["npower"]], axis=1)
        out = okl.project_matrix_diagonal_telescope_to_svd(g[pre + "beam_ut"], svnum, svbounds, npw)
        assert relerr(out, g[pre + "proj_noise"]) < 1e-13


def _sn(g, pre, **kw):
    npw = np.concatenate([g["npower"], g["npower"]], axis=1)
    return okl.sn_covariance(
        g[pre + "beam_svd"], g[pre + "beam_ut"], g[pre + "singularvalues"], g["cv_sg"], g["cv_fg"], npw,
        svcut=float(g["svcut"]), tsys_flat=float(g["tsys_flat"]), **kw
    )


def test_sn_covariance_and_kl(svdkl):
    g = svdkl
    thr = float(g["threshold"])
    for mi in g["mlist"]:
        pre = "m%d_" % mi
        for name, kw in (("kl", {}), ("klnf", dict(use_foregrounds=False))):
            cs, cn = _sn(g, pre, **kw)
            assert relerr(cs, g[pre + name + "_cs"]) < 1e-13
            assert relerr(cn, g[pre + name + "_cn"]) < 1e-13
            evals, evecs, ac = okl.kl_transform_m(cs, cn)
            ref = g[pre + name + "_evals"]
            assert_spectrum(evals, ref, 1e-10, "%s evals m=%d" % (name, mi))
            assert ac == float(g[pre + name + "_ac"])
            full, kept, kvec = okl.threshold_cut(evals, evecs, thr)
            assert kept.size == int(g[pre + name + "_nkept"])
            # E N E^H = I and E S E^H = diag(evals), held to the accuracy the
            # reference's own vectors reach on this (ill-conditioned) pencil
            eref = g[pre + name + "_evecs"]
            scale = max(np.abs(evals).max(), 1e-300)
            rn_ref = relerr(eref @ cn @ eref.T.conj(), np.eye(evals.size), 1.0)
            rs_ref = relerr(eref @ cs @ eref.T.conj(), np.diag(ref), scale)
            assert relerr(evecs @ cn @ evecs.T.conj(), np.eye(evals.size), 1.0) <= max(10 * rn_ref, 1e-8)
            assert relerr(evecs @ cs @ evecs.T.conj(), np.diag(evals), scale) <= max(10 * rs_ref, 1e-8)


def test_doublekl(svdkl):
    g = svdkl
    for mi in g["mlist"]:
        pre = "m%d_" % mi
        evals, evecs, f_evals, ac = okl.doublekl_transform_m(
            lambda th: _sn(g, pre, use_thermal=th), foreground_threshold=float(g["fg_threshold"])
        )
        assert_spectrum(f_evals, g[pre + "dk_f_evals"], 1e-10, "f_evals")
        assert evals.shape == g[pre + "dk_evals"].shape
        assert_spectrum(evals, g[pre + "dk_evals"], 1e-8, "dk evals")
        if evals.size and evals.size < f_evals.size:
            cs, cn = _sn(g, pre, use_thermal=True)
            eref = g[pre + "dk_evecs"]
            rn_ref = relerr(eref @ cn @ eref.T.conj(), np.eye(evals.size), 1.0)
            assert relerr(evecs @ cn @ evecs.T.conj(), np.eye(evals.size), 1.0) <= max(10 * rn_ref, 1e-8)


def _sn_inv(g, pre, use_thermal):
    F, B = int(g["F"]), int(g["B"])
    npw = np.concatenate([g["npower"], g["npower"]], axis=1)
    return okl.sn_covariance(g[pre + "beam_svd"], g[pre + "beam_ut"], g[pre + "singularvalues"], g["cv_sg"], g["cv_fg"],
                             npw, svcut=float(g["svcut"]), use_thermal=use_thermal, tsys_flat=float(g["tsys_flat"]))


def _fix_signs(rows, *others):
    """Real pencils leave only a sign per mode: make the largest component of every row positive."""
    sg = np.array([np.sign(r[np.argmax(np.abs(r))].real) or 1.0 for r in rows])
    return [sg[:, None] * m for m in (rows,) + others]


def test_inverse_and_asymmetric_projection(golden_dir):
    """`inverse = True` (kltransform.py:346-347, doublekl.py:63-67, :83-85) and a frequency-asymmetric sky
    covariance through project_matrix_sky_to_svd, against the unmodified reference on a real-valued pencil."""
    g = np.load(os.path.join(golden_dir, "svdkl_inverse.npz"))
    for mi in g["mlist"]:
        pre = "m%d_" % mi
        svnum, svbounds = osvd.svd_num(g[pre + "singularvalues"], float(g["svcut"]))
        got = okl.project_matrix_sky_to_svd(g[pre + "beam_svd"], svnum, svbounds, g["cv_asym"])
        assert relerr(got, g[pre + "proj_asym"]) < 1e-13
        assert relerr(got, got.T.conj()) > 1e-3   # genuinely not Hermitian: a mirrored lower half would be wrong
        cs, cn = _sn_inv(g, pre, True)
        evals, evecs, ac = okl.kl_transform_m(cs, cn)
        inv = okl.kl_inverse(evecs)
        assert_spectrum(evals, g[pre + "kl_evals"], 1e-10, "kl evals")
        # gauge free: sum over modes of inv[i]^T (x) evecs[i] = E^-1 E = I for both; compare the kept half instead
        k = evals.size // 2
        ours = inv[k:].T @ evecs[k:]
        ref = g[pre + "kl_inv"][k:].T @ g[pre + "kl_evecs"][k:]
        assert relerr(ours, ref, 1.0) < 1e-7
        evals, evecs, f_evals, ac, dinv = okl.doublekl_transform_m(lambda th: _sn_inv(g, pre, th),
                                                                  foreground_threshold=float(g["fg_threshold"]), inverse=True)
        assert evals.shape == g[pre + "dk_evals"].shape and ac == float(g[pre + "dk_ac"])
        assert_spectrum(evals, g[pre + "dk_evals"], 1e-8, "dk evals")
        e1, i1 = _fix_signs(evecs.real, dinv.real)
        e2, i2 = _fix_signs(g[pre + "dk_evecs"].real, g[pre + "dk_inv"].real)
        assert relerr(e1, e2) < 1e-6 and relerr(i1, i2) < 1e-6


def test_eigh_gen(golden_dir):
    g = np.load(os.path.join(golden_dir, "eigh_gen.npz"))
    for case in ("pd", "npd", "zero"):
        ev, evec, ac = okl.eigh_gen(g[case + "_A"], g[case + "_B"])
        assert_spectrum(ev, g[case + "_evals"], 1e-10, case)
        assert np.isclose(ac, float(g[case + "_ac"]), rtol=1e-6, atol=0.0) or ac == float(g[case + "_ac"])
    assert float(g["npd_ac"]) > 0.0


# ---- A17 operators that do not depend on the SVD basis (oracle/projections.py) -----------------
def test_projection_operators_match_reference(golden_dir):
    from oracle import projections as op

    g = np.load(os.path.join(golden_dir, "projections.npz"))
    for tag in ("unpol", "pol"):
        npw, cv = g[tag + "_npower"], g[tag + "_cv"]
        nw0 = npw[0] ** -0.5
        for mi in g[tag + "_mlist"]:
            pre = "%s_m%d_" % (tag, mi)
            bm = g[pre + "beam_m"]

            def rel(a, b):
                return np.abs(a - b).max() / max(np.abs(b).max(), 1e-300)

            # bit-for-bit the same arithmetic as the reference -> 1e-13 is generous
            assert rel(op.invbeam_m(bm, nw0), g[pre + "invbeam_m"]) < 1e-13
            assert rel(op.project_vector_telescope_to_sky(bm, g[pre + "vec_tel"], nw0), g[pre + "tel_to_sky"]) < 1e-13
            assert rel(op.project_vector_backward_dirty(bm, g[pre + "vec_tel"]), g[pre + "backward_dirty"]) < 1e-13
            assert rel(op.project_vector_sky_to_telescope(bm, g[pre + "vec_sky"]), g[pre + "sky_to_tel"]) < 1e-13
            assert rel(op.project_matrix_sky_to_telescope(bm, cv), g[pre + "mat_sky_to_tel"]) < 1e-13
            assert rel(op.project_matrix_sky_to_telescope(bm, cv, True), g[pre + "mat_sky_to_tel_temponly"]) < 1e-13


# ---- G8: exact per-m Fisher matrix (oracle/psfisher.py) ------------------------------------------
def test_psfisher_matches_reference(golden_dir):
    from oracle import psfisher as opf

    g = np.load(os.path.join(golden_dir, "svdkl_unpol.npz"))
    p = np.load(os.path.join(golden_dir, "psfisher.npz"))
    for mi in g["mlist"]:
        pre = "m%d_" % mi
        ev, E = g[pre + "kl_evals"], g[pre + "kl_evecs"]
        i0 = np.searchsorted(ev, float(g["threshold"]))  # transform_save keeps the modes above the KL threshold
        fisher, bias = opf.fisher_m(g[pre + "beam_svd"], g[pre + "svnum"], g[pre + "svbounds"], ev[i0:], E[i0:],
                                    p["clarray"])
        ref = p[pre + "fisher"]
        assert ev[i0:].size == int(p[pre + "nmodes"])
        assert np.abs(fisher - ref).max() <= 1e-13 * np.abs(ref).max()
        assert not bias.any() and not p[pre + "bias"].any()
