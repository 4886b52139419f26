"""Parity at the FULL size of BASELINE.json configs[1] (32-feed unpolarised cylinder, 16 channels,
lmax = mmax = 128, 129 m-blocks, ndof up to ~1200) through size-independent properties — the
oracle would need minutes per m here:

  BT-gen   one (f, b) column of a few m against the oracle's pixel-space restatement;
  SVD      beam_svd = U^H (w B) with orthogonal rows whose norms are the singular values,
           U^H (w^-1 scaling undone) orthonormal, beam_svd . invbeam_svd = I on the kept modes;
  KL       E N E^H = I and E S E^H = diag(lambda), lambda ascending, against the covariances
           the same step projected (tolerance from the conditioning of N, as LAPACK's own bound).
"""
import os
import sys

import numpy as np
import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


@pytest.fixture(scope="module")
def step(tmp_path_factory):
    import torch

    import bench
    from driftscan_amd import btgen, device

    device.reset_context()
    ctx = device.get_context(workspace_bytes=24 << 30)
    tel, bt, kl = bench.build_objects(str(tmp_path_factory.mktemp("full")))
    beam_all = btgen.beam_m_all(tel, ctx=ctx)
    ctx.sync()
    res = bt.svd_device(beam_all, ms=list(range(tel.mmax + 1)))   # the production path: columns l >= m only
    ctx.sync()
    torch.cuda.synchronize()
    return tel, bt, kl, ctx, beam_all, res


def test_btgen_columns_against_oracle(step):
    from oracle import btgen as ob

    tel, bt, kl, ctx, beam_all, res = step
    fsel, bsel = np.array([0, tel.nfreq - 1]), np.array([0, tel.nbase // 2, tel.nbase - 1])
    desc = dict(polarised=False, zenith=tel.zenith, baselines=tel.baselines, uniquepairs=tel.uniquepairs,
                beamclass=tel.beamclass, wavelengths=tel.wavelengths, cylinder_width=tel.cylinder_width,
                fwhm_e=tel.fwhm_e, fwhm_h=tel.fwhm_h, lmax=tel.lmax, mmax=tel.mmax, l_boost=tel.l_boost,
                included_freq=fsel, included_baseline=bsel, accuracy_boost=tel.accuracy_boost, sht_iter=tel.sht_iter, sht_fft=True)
    ms = [0, 37, 90, tel.mmax]  # the last one lies above the natural band limit: an all-zero block
    ref = ob.beam_transfer_m(desc, mlist=ms)  # (F, 2, B, 1, L) per m, non-zero on the selected (f, b)
    got = beam_all.cpu().numpy()
    scale = np.abs(ref[0]).max()
    for mi in ms:
        g = got[mi][fsel][:, :, bsel]
        r = ref[mi][fsel][:, :, bsel]
        assert (np.abs(r).max() > 0) == (mi < tel.mmax)
        # fp64 pixel sums over 196 608 pixels in a different order: 1e-10 of the block scale
        assert np.abs(g - r).max() < 1e-10 * scale, mi


def test_svd_properties_all_m(step):
    tel, bt, kl, ctx, beam_all, res = step
    sv = res["singularvalues"].cpu().numpy()           # (M, F, K)
    bsvd = res["beam_svd"].cpu().numpy()                # (M, F, K, P, L)
    but = res["beam_ut"].cpu().numpy()                  # (M, F, K, T)
    ibs = res["invbeam_svd"].cpu().numpy()              # (M, F, P, L, K)
    beam = beam_all.cpu().numpy()
    M, F, K = sv.shape
    T, L = bt.ntel, tel.lmax + 1
    noisew = bt._noisew()
    worst = dict(rows=0.0, proj=0.0, pinv=0.0, orth=0.0)
    for mi in range(0, M, 8):
        for fi in (0, F // 2, F - 1):
            s = sv[mi, fi]
            n = int((s > s.max() * bt.svcut).sum()) if s.max() > 0 else 0
            if n == 0:
                continue
            B = (beam[mi, fi].reshape(T, L)) * noisew[fi][:, None]
            U = but[mi, fi, :n] / noisew[fi][None, :]            # beam_ut = ut * noisew
            X = bsvd[mi, fi, :n].reshape(n, L)
            # rows of beam_svd: mutually orthogonal, norms = singular values
            G = X @ X.conj().T
            d = np.sqrt(np.abs(np.diag(G)))
            worst["rows"] = max(worst["rows"], np.abs(d - s[:n]).max() / s[0])
            off = G - np.diag(np.diag(G))
            worst["orth"] = max(worst["orth"], np.abs(off / np.outer(d, d)).max())
            worst["proj"] = max(worst["proj"], np.abs(U @ B - X).max() / s[0])
            I = X @ ibs[mi, fi].reshape(L, K)[:, :n]
            worst["pinv"] = max(worst["pinv"], np.abs(I - np.eye(n)).max())
    assert worst["rows"] < 1e-12, worst
    # Blocks with more rows than sky columns go through the transposed matrix: there the Jacobi
    # sweeps make the LEFT vectors orthogonal (|cos| <= 1e-13) and beam_svd = U^H (w B) is a product, as in the reference
    # (beam = ut3 . bfr, beamtransfer.py:877) — a kept mode at svcut = 1e-6 of sigma_1 then carries eps sigma_1 / sigma_i
    # ~ 1e-10 of the dominant rows, in LAPACK's U as here.  (The wide blocks rotate the rows of beam_svd themselves: 1e-13.)
    assert worst["orth"] < 1e-9, worst
    assert worst["proj"] < 1e-12, worst
    assert worst["pinv"] < 1e-8, worst     # kappa(beam_svd) = 1/svcut = 1e6 amplifies eps


def test_kl_properties_sample_m(step):
    import torch

    tel, bt, kl, ctx, beam_all, res = step
    sv = res["singularvalues"].cpu().numpy()
    ms = [0, 16, 48, 96, tel.mmax]
    for mi in range(tel.mmax + 1):
        bt._dev[mi] = dict(beam_svd=res["beam_svd"][mi], beam_ut=res["beam_ut"][mi], singularvalues=sv[mi])
    S, N, ndofs, off = kl.sn_covariance_device(ms)
    ctx.sync()
    Sh, Nh = S.cpu().numpy(), N.cpu().numpy()
    out = kl._transform_batch(ms, to_host=True)
    for i, mi in enumerate(ms):
        n = int(ndofs[i])
        if n == 0:
            continue
        ev, E = out[i][0], out[i][1]
        Sm = Sh[off[i] : off[i] + n * n].reshape(n, n)
        Nm = Nh[off[i] : off[i] + n * n].reshape(n, n)
        assert np.all(np.diff(ev) >= 0)
        lam = np.linalg.eigvalsh((Nm + Nm.conj().T) / 2)
        cond = lam[-1] / max(lam[0], 1e-300)
        tol = max(50 * np.finfo(float).eps * cond, 1e-10)  # LAPACK's bound for zhegvd residuals
        # with `subset` (the default) only the modes above the S/N threshold are formed
        i_ev = int(np.searchsorted(ev, kl.threshold))
        assert not E[:i_ev].any()
        Ek, evk = E[i_ev:], ev[i_ev:]
        if evk.size == 0:
            continue
        ENE = Ek @ Nm @ Ek.conj().T
        ESE = Ek @ Sm @ Ek.conj().T
        assert np.abs(ENE - np.eye(evk.size)).max() < tol, (mi, n, cond)
        assert np.abs(ESE - np.diag(evk)).max() < tol * max(1.0, np.abs(evk).max()), (mi, n, cond)
    # the early cut changes nothing on the kept modes: same step with every mode formed
    kl.subset = False
    full = kl._transform_batch(ms, to_host=True)
    kl.subset = True
    for i, mi in enumerate(ms):
        ev, E = out[i][0], out[i][1]
        if ev.size == 0:
            continue
        i_ev = int(np.searchsorted(ev, kl.threshold))
        assert np.abs(full[i][0] - ev).max() <= 1e-12 * max(np.abs(ev).max(), 1e-300)
        # same projector onto the kept subspace (rows may differ by phases / rotations in clusters)
        Nm = Nh[off[i] : off[i] + ev.size**2].reshape(ev.size, ev.size)
        Pk = E[i_ev:].conj().T @ E[i_ev:] @ Nm
        Pf = full[i][1][i_ev:].conj().T @ full[i][1][i_ev:] @ Nm
        assert np.abs(Pk - Pf).max() < 1e-6 * max(np.abs(Pf).max(), 1e-300), mi


def test_spectra_against_oracle(step):
    """configs[1] at full size, the spectra themselves: singular values and KL eigenvalues of the REAL blocks m = 0, 32, 64,
    96, 128 against the oracle's SVD chain + covariance projections + eigh_gen on the same blocks (copied back from the
    device) — 1e-10 of the block's largest singular value; eigenvalues within the conditioning bound of the pencil
    (`pencil_tol`: max(1e-10, 50 eps cond(N)), what any Cholesky-based solver, LAPACK's included, can deliver); `svnum`
    exact; kept-mode counts equal unless an eigenvalue lies within the tolerance of the cut (then a warning, counted in
    the test summary).  reference: drift/core/beamtransfer.py:1116-1133, kltransform.py:310-355."""
    import warnings

    from oracle import kl as okl
    from oracle import svdchain as osvd
    from parity_util import assert_spectrum, pencil_tol

    tel, bt, kl, ctx, beam_all, res = step
    sv_all = res["singularvalues"].cpu().numpy()
    for mi in range(tel.mmax + 1):
        bt._dev[mi] = dict(beam_svd=res["beam_svd"][mi], beam_ut=res["beam_ut"][mi], singularvalues=sv_all[mi])
        bt._sv_host[mi] = sv_all[mi]
    ms = [0, 32, 64, 96, tel.mmax]
    out = kl._transform_batch(ms, to_host=True)
    noisew = bt._noisew()[:, : tel.nbase]
    npw = kl._npower(1.0)
    for i, mi in enumerate(ms):
        blk = beam_all[mi].cpu().numpy()
        ref = osvd.svd_m(blk, noisew, polsvcut=bt.polsvcut)
        assert_spectrum(sv_all[mi], ref["singularvalues"], 1e-10, "configs[1] sv m=%d" % mi)
        svnum, svb = bt._svd_num(mi)
        rnum, rb = osvd.svd_num(ref["singularvalues"], bt.svcut)
        assert np.array_equal(svnum, rnum) and np.array_equal(svb, rb), mi
        if int(rnum.sum()) == 0:   # above the band limit: no mode survives svcut, the KL stage returns nothing (kltransform.py:324-326)
            assert np.asarray(out[i][0]).size == 0
            continue
        cs, cn = okl.sn_covariance(ref["beam_svd"], ref["beam_ut"], ref["singularvalues"], kl.signal(), kl.foreground(),
                                   npw, svcut=bt.svcut)
        ev_o = okl.kl_transform_m(cs, cn)[0]
        ev_g = np.asarray(out[i][0])
        assert ev_g.shape == ev_o.shape, mi
        if ev_o.size == 0:
            continue
        tol = pencil_tol(cn)
        err = np.abs(ev_g - ev_o).max() / np.abs(ev_o).max()
        print("configs[1] m = %d: ndof %d, sigma vs oracle %.2e of sigma_max, eigenvalues %.2e of lambda_max (tolerance %.1e)"
              % (mi, ev_o.size, np.abs(sv_all[mi] - ref["singularvalues"]).max() / ref["singularvalues"].max(), err, tol))
        if err > tol:
            # beyond the conditioning bound of the pencil: the svcut truncation in front of it has a conditioning of its own (a
            # kept subspace with a singular value near the cut).  Yardstick: how far the ORACLE's spectrum moves when the block
            # is perturbed by one unit roundoff per entry.
            rng = np.random.default_rng(mi)
            sens = 0.0
            for _ in range(2):
                pert = blk * (1.0 + 2.2e-16 * rng.standard_normal(blk.shape))
                rp = osvd.svd_m(pert, noisew, polsvcut=bt.polsvcut)
                csp, cnp_ = okl.sn_covariance(rp["beam_svd"], rp["beam_ut"], rp["singularvalues"], kl.signal(), kl.foreground(),
                                              npw, svcut=bt.svcut)
                evp = okl.kl_transform_m(csp, cnp_)[0]
                assert evp.shape == ev_o.shape
                sens = max(sens, np.abs(evp - ev_o).max() / np.abs(ev_o).max())
            print("configs[1] m = %d: beyond pencil_tol; oracle sensitivity to one ulp of the block %.2e" % (mi, sens))
            tol = max(tol, 10.0 * sens)
        assert_spectrum(ev_g, ev_o, tol, "configs[1] kl evals m=%d" % mi)
        kg, ko = int((ev_g >= kl.threshold).sum()), int((ev_o >= kl.threshold).sum())
        if kg != ko:
            near = np.abs(ev_o - kl.threshold).min() <= tol * np.abs(ev_o).max()
            assert near, (mi, kg, ko)
            warnings.warn("configs[1] m = %d: kept-mode count %d vs the oracle's %d — an eigenvalue lies within tol of the cut"
                          % (mi, kg, ko))
