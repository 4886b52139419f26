"""GPU parity of the operator classes (BeamTransfer / KLTransform / DoubleKL /
ProductManager) against golden vectors of the unmodified reference and against the
numpy oracle, going through the same method names the reference's tests use."""
import os

import numpy as np
import pytest

from parity_util import assert_spectrum, pencil_sensitivity, pencil_tol, relerr

pytestmark = pytest.mark.gpu


class FakeTelescope(object):
    """The attribute surface BeamTransfer/KLTransform read."""

    def __init__(self, g):
        self.nfreq, self.nbase = int(g["F"]), int(g["B"])
        self.npairs = self.nbase
        self.num_pol_sky = int(g["P"])
        self.lmax = self.mmax = int(g["lmax"])
        self.included_freq = np.arange(self.nfreq)
        self.included_baseline = np.arange(self.nbase)
        self.included_pol = np.arange(self.num_pol_sky)
        self.frequencies = g["frequencies"]
        self.baselines = np.zeros((self.nbase, 2))
        self.tsys_flat = float(g["tsys_flat"])
        self._npower = g["npower"]

    def noisepower(self, bl_indices, f_indices, ndays=None):
        bl, fi = np.broadcast_arrays(bl_indices, f_indices)
        return self._npower[fi, bl]


@pytest.fixture(scope="module", params=["unpol", "pol"])
def setup(request, golden_dir, tmp_path_factory):
    from driftscan_amd import beamtransfer, device, storage

    device.reset_context()
    g = np.load(os.path.join(golden_dir, "svdkl_%s.npz" % request.param))
    tel = FakeTelescope(g)
    d = str(tmp_path_factory.mktemp("bt_" + request.param))
    bt = beamtransfer.BeamTransfer(d, telescope=tel)
    bt.polsvcut = float(g["polsvcut"])
    bt.svcut = float(g["svcut"])
    bt._generate_dirs()
    mlist = [int(m) for m in g["mlist"]]
    for mi in mlist:
        with storage.File(bt._mfile(mi), "w") as f:
            f.create_dataset("beam_m", data=g["m%d_beam_m" % mi][..., mi:])
    bt._my_ms = lambda mlist_=None: mlist  # only the fixture's m exist
    bt._generate_svdfiles(regen=True)
    return g, bt, mlist


def test_beamtransfer_svd_products(setup):
    g, bt, mlist = setup
    for mi in mlist:
        pre = "m%d_" % mi
        assert np.array_equal(bt.beam_m(mi), g[pre + "beam_m"])
        assert_spectrum(bt.beam_singularvalues(mi), g[pre + "singularvalues"], 1e-10, "sv m=%d" % mi)
        svnum, svbounds = bt._svd_num(mi)
        assert np.array_equal(svnum, g[pre + "svnum"]) and np.array_equal(svbounds, g[pre + "svbounds"])
        assert bt.ndof(mi) == g[pre + "svbounds"][-1]
        assert bt.beam_svd(mi).shape == g[pre + "beam_svd"].shape
        assert bt.invbeam_svd(mi).shape == g[pre + "invbeam_svd"].shape
        assert bt.beam_ut(mi).shape == g[pre + "beam_ut"].shape
        # projections of the reference's C_l through *our* SVD products: gauge invariant
        for key, cv, to in (("proj_sg", "cv_sg", False), ("proj_fg", "cv_fg", False), ("proj_sg_temponly", "cv_sg", True)):
            ours = bt.project_matrix_sky_to_svd(mi, g[cv], temponly=to)
            ref = g[pre + key]
            assert ours.shape == ref.shape
            assert_spectrum(np.linalg.eigvalsh(ours), np.linalg.eigvalsh(ref), 1e-9, key)
    assert bt.svd_all().shape == (bt.telescope.mmax + 1, bt.nfreq, bt.svd_len)


def _kl_objects(g, bt):
    from driftscan_amd import doublekl, kltransform

    out = {}
    for name, klass, kw in (("kl", kltransform.KLTransform, {}), ("klnf", kltransform.KLTransform, dict(use_foregrounds=False)),
                            ("dk", doublekl.DoubleKL, dict(foreground_threshold=float(g["fg_threshold"])))):
        obj = klass.from_config(dict(threshold=float(g["threshold"]), **kw), bt, subdir=name)
        obj._cvsg, obj._cvfg = g["cv_sg"], g["cv_fg"]
        out[name] = obj
    return out


def test_kltransform(setup):
    g, bt, mlist = setup
    kls = _kl_objects(g, bt)
    for name in ("kl", "klnf"):
        kl = kls[name]
        kl.generate(regen=True)
        for mi in mlist:
            pre = "m%d_%s_" % (mi, name)
            cs, cn = g[pre + "cs"], g[pre + "cn"]
            tol = pencil_tol(cn)
            if tol > 1e-6:
                tol = max(1e-10, 10 * pencil_sensitivity(cs, cn))
            ref = g[pre + "evals"]
            from driftscan_amd import storage

            with storage.File(kl._evfile % mi, "r") as f:
                full = f["evals_full"][:]
                kept = f["evals"][:]
                evecs = f["evecs"][:]
                assert f.attrs["FLAGS"] == "Normal"
            assert_spectrum(full, ref, tol, "%s evals m=%d" % (name, mi))
            assert kept.size == int(g[pre + "nkept"])
            assert evecs.shape == (kept.size, ref.size)
            ev, evec = kl.modes_m(mi)
            if kept.size:
                assert np.array_equal(ev, kept)
                # E N E^H = I on the kept modes, in the (gauge-dependent) basis of OUR svd products
                s_ours, n_ours = kl.sn_covariance(mi)
                assert relerr(evec @ n_ours @ evec.T.conj(), np.eye(kept.size), 1.0) < max(1e-8, 100 * tol)
                assert relerr(evec @ s_ours @ evec.T.conj(), np.diag(kept), np.abs(ref).max()) < max(1e-8, 100 * tol)
        assert kl.evals_all().shape == (bt.telescope.mmax + 1, bt.ndofmax)


def test_doublekl(setup):
    g, bt, mlist = setup
    from driftscan_amd import storage

    dk = _kl_objects(g, bt)["dk"]
    dk.generate(regen=True)
    for mi in mlist:
        pre = "m%d_dk_" % mi
        tol1 = max(1e-10, 10 * pencil_sensitivity(g[pre + "cs_nothermal"], g[pre + "cn_nothermal"]))
        with storage.File(dk._evfile % mi, "r") as f:
            f_evals = f["f_evals"][:]
            full = f["evals_full"][:]
            kept = f["evals"][:]
            evecs = f["evecs"][:]
            flags = f.attrs["FLAGS"]
        assert_spectrum(f_evals, g[pre + "f_evals"], tol1, "f_evals m=%d" % mi)
        ref = g[pre + "evals"]
        # the number of modes passing the foreground cut is exact (doublekl.py:56-60) ...
        assert int((f_evals > dk.foreground_threshold).sum()) == ref.size
        assert int((g[pre + "f_evals"] > dk.foreground_threshold).sum()) == ref.size
        assert not full[: full.size - ref.size].any()          # evals_full is right aligned, zero padded
        got = full[full.size - ref.size :]
        # ... then the stage-2 spectrum and the S/N cut transform_save applies to it (kltransform.py:388-398)
        if ref.size:
            tol2 = max(1e-8, 100 * tol1)
            assert_spectrum(got, ref, tol2, "dk evals m=%d" % mi)
            err = np.abs(got - ref).max() / np.abs(ref).max()
            print("DoubleKL m=%d: stage-2 spectrum error %.2e of lambda_max (bound %.1e)" % (mi, err, tol2))
            nkept_ref = ref.size - int(np.searchsorted(ref, dk.threshold))
            near = np.abs(ref - dk.threshold).min() <= tol2 * np.abs(ref).max()
            assert kept.size == nkept_ref or near
            assert evecs.shape == (kept.size, f_evals.size)
        # `ac`/FLAGS come from stage 1 (doublekl.py:58): the fixture's stage-1 pencils are positive definite
        assert (flags == "Normal") == (float(g[pre + "ac"]) == 0.0)


def test_one_call_c_drivers_match_the_classes(setup):
    """dm_kl_m / dm_doublekl_m (the one-call drivers of include/driftmi.h for hosts without the Python layer) against
    KLTransform / DoubleKL, which compose the same entry points one by one."""
    import torch

    from driftscan_amd import device
    from driftscan_amd.beamtransfer import BeamTransfer

    g, bt, mlist = setup
    kls = _kl_objects(g, bt)
    ctx = device.get_context()
    prods = [bt._dev_products(mi) for mi in mlist]
    bsvd = torch.stack([p["beam_svd"] for p in prods])
    but = torch.stack([p["beam_ut"] for p in prods])
    svnum = np.stack([bt._svd_num(mi)[0] for mi in mlist])
    kl, dk = kls["kl"], kls["dk"]
    cl_sg, m_sg, s_sg = BeamTransfer._cl_device(kl.signal())
    cl_fg, m_fg, s_fg = BeamTransfer._cl_device(kl.foreground())
    npw = ctx.to_device(kl._npower(1.0))
    # ---- KLTransform
    ref = kl._transform_batch(mlist, to_host=True)
    ev, evoff, E, off, ac, nk = ctx.kl_m(bsvd, but, svnum, np.array(mlist), cl_sg, m_sg, s_sg, cl_fg, m_fg, s_fg, npw, 1.0,
                                         kl._foreground_regulariser, cut=("upper", kl.threshold))
    evh, Eh = ev.cpu().numpy(), E.cpu().numpy()
    for i, mi in enumerate(mlist):
        n = int(svnum[i].sum())
        assert np.abs(evh[evoff[i] : evoff[i] + n] - ref[i][0]).max() <= 1e-12 * np.abs(ref[i][0]).max()
        assert np.abs(Eh[off[i] : off[i] + n * n].reshape(n, n) - ref[i][1]).max() <= 1e-9 * np.abs(ref[i][1]).max()
        assert int(nk[i]) == n - int(np.searchsorted(ref[i][0], kl.threshold)) and ac[i] == ref[i][3]["ac"]
    # ---- DoubleKL
    refd = dk._transform_batch(mlist, to_host=True)
    nc1 = (1e-3 / bt.telescope.tsys_flat) ** 2
    fev, ev2, evoff, M, off, nm, nk2, ac1 = ctx.doublekl_m(bsvd, but, svnum, np.array(mlist), cl_sg, m_sg, s_sg, cl_fg, m_fg, s_fg,
                                                          npw, nc1, dk._foreground_regulariser, dk.foreground_threshold,
                                                          cut=("upper", dk.threshold))
    fevh, ev2h, Mh = fev.cpu().numpy(), ev2.cpu().numpy(), M.cpu().numpy()
    for i, mi in enumerate(mlist):
        n = int(svnum[i].sum())
        e_ref, m_ref, _, ex = refd[i]
        assert int(nm[i]) == e_ref.size and ac1[i] == ex["ac"]
        # the driver scales the noise term inside the product (alpha * U d U^H), the class scales d first: one rounding
        # apart in N, which the stage-1 pencil (cond ~ 1e5) amplifies to ~1e-12 of lambda_max
        assert np.abs(fevh[evoff[i] : evoff[i] + n] - ex["f_evals"]).max() <= 1e-9 * np.abs(ex["f_evals"]).max()
        if e_ref.size:
            r = e_ref.size
            assert np.abs(ev2h[evoff[i] : evoff[i] + r] - e_ref).max() <= 1e-8 * np.abs(e_ref).max()
            # rows are defined up to a phase: compare the gauge-free products M^H diag M
            Mo = Mh[off[i] : off[i] + r * n].reshape(r, n)
            k0 = int(np.searchsorted(e_ref, dk.threshold))
            Po, Pr = Mo[k0:].conj().T @ Mo[k0:], m_ref[k0:].conj().T @ m_ref[k0:]
            assert np.abs(Po - Pr).max() <= 1e-6 * max(np.abs(Pr).max(), 1e-300)


def test_inverse_and_asymmetric_covariance(golden_dir, tmp_path):
    """`inverse: Yes` for KLTransform and DoubleKL on the device (N E^H instead of an LU inversion per m,
    kltransform.py:124-143, :346-347; doublekl.py:63-67, :83-85) and a sky covariance that is NOT symmetric
    under f <-> f' through project_matrix_sky_to_svd (beamtransfer.py:1135-1188 takes any array)."""
    from driftscan_amd import beamtransfer, device, doublekl, kltransform, storage

    device.reset_context()
    g = np.load(os.path.join(golden_dir, "svdkl_inverse.npz"))
    tel = FakeTelescope(g)
    bt = beamtransfer.BeamTransfer(str(tmp_path / "bt"), telescope=tel)
    bt.polsvcut, bt.svcut = float(g["polsvcut"]), float(g["svcut"])
    bt._generate_dirs()
    mlist = [int(m) for m in g["mlist"]]
    for mi in mlist:
        with storage.File(bt._mfile(mi), "w") as f:
            f.create_dataset("beam_m", data=g["m%d_beam_m" % mi][..., mi:])
    bt._my_ms = lambda mlist_=None: mlist
    bt._generate_svdfiles(regen=True)
    kl = kltransform.KLTransform.from_config(dict(threshold=float(g["threshold"]), inverse=True), bt, subdir="kli")
    dk = doublekl.DoubleKL.from_config(dict(threshold=float(g["threshold"]), inverse=True,
                                            foreground_threshold=float(g["fg_threshold"])), bt, subdir="dki")
    for o in (kl, dk):
        o._cvsg, o._cvfg = g["cv_sg"], g["cv_fg"]
        o.generate(regen=True)

    for mi in mlist:
        pre = "m%d_" % mi
        # the asymmetric covariance: compare through OUR svd basis (gauge: U -> D U per frequency leaves
        # the singular values of the non-Hermitian projection unchanged)
        ours = bt.project_matrix_sky_to_svd(mi, g["cv_asym"])
        ref = g[pre + "proj_asym"]
        assert np.abs(ours - ours.T.conj()).max() > 1e-3 * np.abs(ours).max()
        sv_o, sv_r = np.linalg.svd(ours, compute_uv=False), np.linalg.svd(ref, compute_uv=False)
        assert np.abs(sv_o - sv_r).max() < 1e-9 * sv_r.max()
        assert np.abs(np.linalg.eigvals(ours).real.sum() - np.trace(ref).real) < 1e-9 * np.abs(ref).max() * ref.shape[0]
        # KLTransform inverse: evinv^T is the inverse of the mode matrix
        with storage.File(kl._evfile % mi, "r") as f:
            ev, E, inv = f["evals"][:], f["evecs"][:], f["evinv"][:]
        i0 = int(np.searchsorted(g[pre + "kl_evals"], kl.threshold))
        assert ev.size == g[pre + "kl_evals"].size - i0
        assert_spectrum(ev, g[pre + "kl_evals"][i0:], 1e-9, "kl evals (inverse)")
        assert inv.shape == E.shape
        assert np.abs(E @ inv.T - np.eye(ev.size)).max() < 1e-8
        # against the reference: the modes live in the coordinates of the SVD basis, whose rows carry a gauge of
        # their own (on this real problem a sign per SVD mode and a sign per KL mode).  inv^T E is the projector
        # onto the kept modes in those coordinates: it changes as D (inv^T E) D under the coordinate signs D, so
        # its element-wise modulus is what the two implementations must share.
        refP = g[pre + "kl_inv"][i0:].T @ g[pre + "kl_evecs"][i0:]
        assert np.abs(np.abs(inv.T @ E) - np.abs(refP)).max() < 1e-7 * max(np.abs(refP).max(), 1.0)
        assert np.array_equal(kl.invmodes_m(mi), inv.T)
        # DoubleKL inverse: the reference's formula, unique on this real pencil up to a sign per mode
        with storage.File(dk._evfile % mi, "r") as f:
            ev, E, inv = f["evals"][:], f["evecs"][:], f["evinv"][:]
        rE, rI, rv = g[pre + "dk_evecs"], g[pre + "dk_inv"], g[pre + "dk_evals"]
        j0 = int(np.searchsorted(rv, dk.threshold))
        assert ev.size == rv.size - j0 and E.shape == inv.shape == (ev.size, rE.shape[1])
        assert np.abs(E.imag).max() < 1e-9 * np.abs(E).max() and np.abs(inv.imag).max() < 1e-9 * np.abs(inv).max()
        # on a real pencil the reference's inv2 @ inv1 is a true inverse of the composed modes ...
        assert np.abs(E @ inv.T - np.eye(ev.size)).max() < 1e-7
        assert np.abs(rE[j0:] @ rI[j0:].T - np.eye(ev.size)).max() < 1e-7
        # ... and the same projector, up to the coordinate signs of the SVD basis (see above)
        refP = rI[j0:].T @ rE[j0:]
        assert np.abs(np.abs(inv.T @ E) - np.abs(refP)).max() < 1e-6 * max(np.abs(refP).max(), 1.0)
        # row-wise: |modes| and |inverse rows| agree once the SVD-basis signs are removed by the modulus
        assert np.abs(np.abs(E) - np.abs(rE[j0:])).max() < 1e-6 * np.abs(rE).max()
        assert np.abs(np.abs(inv) - np.abs(rI[j0:])).max() < 1e-6 * np.abs(rI).max()


def test_product_manager_end_to_end(tmp_path):
    """ProductManager.from_config -> generate() on a small cylinder, compared with the oracle chain."""
    import yaml

    from driftscan_amd import device, manager, storage
    from oracle import btgen as ob
    from oracle import kl as okl
    from oracle import svdchain as osvd

    device.reset_context()
    conf = dict(
        config=dict(beamtransfers=True, kltransform=True, psfisher=True, output_directory=str(tmp_path / "prod"),
                    polsvcut=1e-4, truncate=False),
        psfisher=[dict(type="Full", name="ps", klname="kl", threshold=0.0, bandtype="polar", num_theta=1,
                       k_bands=[dict(spacing="linear", start=0.0, stop=0.006, num=4)])],
        telescope=dict(type="UnpolarisedCylinder", num_freq=3, freq_start=400.0, freq_end=430.0, freq_mode="edge",
                       num_cylinders=2, cylinder_width=2.0, num_feeds=3, feed_spacing=0.4, tsys=1.0),
        kltransform=[dict(type="KLTransform", name="kl", use_foregrounds=False),
                     dict(type="DoubleKL", name="dk", foreground_threshold=10.0)],
    )
    cfile = tmp_path / "params.yaml"
    cfile.write_text(yaml.dump(conf))
    pm = manager.ProductManager.from_config(str(cfile))
    pm.generate()
    t, bt = pm.telescope, pm.beamtransfer
    assert os.path.exists(bt.directory + "/beam_m/COMPLETED")
    # oracle chain on the same telescope
    desc = dict(polarised=False, zenith=t.zenith, baselines=t.baselines, uniquepairs=t.uniquepairs,
                beamclass=t.beamclass, wavelengths=t.wavelengths, cylinder_width=t.cylinder_width, fwhm_e=t.fwhm_e,
                fwhm_h=t.fwhm_h, lmax=t.lmax, mmax=t.mmax, l_boost=t.l_boost, included_freq=t.included_freq,
                included_baseline=t.included_baseline, accuracy_boost=t.accuracy_boost, sht_iter=t.sht_iter, sht_fft=True)
    ms = [0, 5, t.mmax]
    ref_bm = ob.beam_transfer_m(desc, mlist=ms)
    noisew = bt._noisew()[:, : t.nbase]
    kl = pm.kltransforms["kl"]
    npw = kl._npower(1.0)
    for mi in ms:
        bm = bt.beam_m(mi)
        assert np.abs(bm - ref_bm[mi]).max() < 1e-9 * np.abs(ref_bm[0]).max()
        ref_svd = osvd.svd_m(ref_bm[mi], noisew, polsvcut=bt.polsvcut)
        assert_spectrum(bt.beam_singularvalues(mi), ref_svd["singularvalues"], 1e-9, "sv")
        cs, cn = okl.sn_covariance(ref_svd["beam_svd"], ref_svd["beam_ut"], ref_svd["singularvalues"], kl.signal(),
                                   kl.foreground(), npw, svcut=bt.svcut, use_foregrounds=False)
        ev, _, _ = okl.kl_transform_m(cs, cn)
        with storage.File(kl._evfile % mi, "r") as f:
            assert_spectrum(f["evals_full"][:], ev, 1e-8, "kl evals m=%d" % mi)
    assert pm.kltransforms["dk"].evals_all().shape == (t.mmax + 1, bt.ndofmax)
    # Fisher matrix of the (stand-in) multipole bands: the file of the manager run against the
    # oracle chain summed over every m
    from driftscan_amd import psestimation
    from oracle import psfisher as opf

    ps = pm.psestimators["ps"]
    fisher, bias = ps.fisher_bias()
    kb = np.linspace(0.0, 0.006, 4)
    clarray = psestimation.band_clarray_standin(t.lmax, t.frequencies, kb[:-1], kb[1:])
    assert clarray.shape[0] == 3 and np.abs(clarray).max() > 0
    ref = np.zeros((3, 3))
    for mi in range(t.mmax + 1):
        ev, E = kl.modes_m(mi, threshold=0.0)
        if ev is None:
            continue
        svnum, svb = bt._svd_num(mi)
        ref += opf.fisher_m(bt.beam_svd(mi), svnum, svb, ev, E, clarray)[0].real
    assert np.abs(ref).max() > 0
    assert relerr(fisher, ref) < 1e-8
    assert not bias.any()


def test_generate_mfiles_in_m_ranges(tmp_path):
    """beam_m generation with the blocks of one call bounded (`beam_chunk_gb`): several m sub-ranges, each
    with its own map synthesis, give the files of the all-m call bit for bit (BASELINE configs[2] cannot hold
    its 934 GB of beam_m blocks in one call)."""
    from driftscan_amd import beamtransfer, cylinder, device, storage

    device.reset_context()
    tcfg = dict(num_freq=2, freq_start=400.0, freq_end=420.0, freq_mode="edge", num_cylinders=2, cylinder_width=2.0,
                num_feeds=3, feed_spacing=0.4, tsys=1.0)
    out = {}
    for tag, gb in (("all", 96.0), ("ranges", None)):
        tel = cylinder.PolarisedCylinderTelescope.from_config(dict(tcfg))
        bt = beamtransfer.BeamTransfer(str(tmp_path / tag), telescope=tel)
        if gb is None:
            per_block = tel.nfreq * 2 * tel.nbase * tel.num_pol_sky * (tel.lmax + 1) * 16
            bt.beam_chunk_gb = 5.5 * per_block / (1 << 30)  # five blocks per call
        bt._generate_dirs()
        bt._generate_mfiles(regen=True)
        assert (bt._beam_all is not None) == (tag == "all")
        out[tag] = [bt.beam_m(mi) for mi in range(tel.mmax + 1)]
    assert len(out["all"]) > 11
    for a, b in zip(out["all"], out["ranges"]):
        assert np.array_equal(a, b)


_RANK_SCRIPT = r'''
import os, sys
sys.path.insert(0, %(root)r)
import torch.distributed as dist
dist.init_process_group(backend="gloo", init_method="tcp://127.0.0.1:%(port)d", rank=int(sys.argv[1]), world_size=2)
from driftscan_amd import manager
pm = manager.ProductManager.from_config(%(cfile)r)
pm.generate()
dist.barrier()
dist.destroy_process_group()
'''


def test_two_ranks_on_one_gpu_match_single_process(tmp_path):
    """ProductManager.generate() with two ranks (gloo between the processes, both on this GPU, sharing the
    output directory as MPI ranks of the reference do) against the single-process run: every rank takes ONE contiguous,
    cost-balanced range of m through BT-gen -> SVD -> KL while the blocks are resident, gathered spectra, all-reduced
    Fisher matrix.  The beam_m files of the two runs are byte-identical and no rank reads one back."""
    import socket
    import subprocess
    import sys

    import yaml

    from driftscan_amd import device, manager, storage

    def conf(outdir):
        return dict(
            config=dict(beamtransfers=True, kltransform=True, psfisher=True, output_directory=str(outdir), polsvcut=1e-4,
                        truncate=False),
            psfisher=[dict(type="Full", name="ps", klname="kl", threshold=0.0, bandtype="polar", num_theta=1,
                           k_bands=[dict(spacing="linear", start=0.0, stop=0.006, num=4)])],
            telescope=dict(type="PolarisedCylinder", num_freq=3, freq_start=400.0, freq_end=430.0, freq_mode="edge",
                           num_cylinders=2, cylinder_width=2.0, num_feeds=3, feed_spacing=0.4, tsys=1.0),
            kltransform=[dict(type="KLTransform", name="kl", use_foregrounds=True, threshold=0.0),
                         dict(type="DoubleKL", name="dk", foreground_threshold=10.0)],
        )

    device.reset_context()
    c1 = tmp_path / "one.yaml"
    c1.write_text(yaml.dump(conf(tmp_path / "one")))
    pm = manager.ProductManager.from_config(str(c1))
    pm.generate()
    device.reset_context()

    c2 = tmp_path / "two.yaml"
    c2.write_text(yaml.dump(conf(tmp_path / "two")))
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    script = tmp_path / "rank.py"
    script.write_text(_RANK_SCRIPT % dict(root=os.path.dirname(os.path.dirname(os.path.abspath(__file__))), port=port,
                                          cfile=str(c2)))
    trace = tmp_path / "opens.txt"
    env = dict(os.environ, DRIFTMI_DEVICE="0", DRIFTMI_WORKSPACE_GB="2", DRIFTMI_TRACE_OPEN=str(trace))
    procs = [subprocess.Popen([sys.executable, str(script), str(r)], env=env, stdout=subprocess.PIPE,
                              stderr=subprocess.STDOUT) for r in range(2)]
    outs = [p.communicate(timeout=300)[0].decode() for p in procs]
    assert all(p.returncode == 0 for p in procs), "\n".join(outs)

    # the blocks of a rank go from generation through the SVD chain to the KL stage while they are resident: no beam_m
    # file and no svd file is opened for reading before every product of the run has been written
    lines = trace.read_text().splitlines()
    first_ps = next(i for i, l in enumerate(lines) if l.startswith("#"))   # both ranks are past the KL barrier here
    opens = [l.split(" ", 1) for l in lines if not l.startswith("#")]
    first_ps -= sum(1 for l in lines[:first_ps] if l.startswith("#"))
    assert any(mode == "w" and "/beam_m/" in path for mode, path in opens)
    assert any(mode == "w" and path.endswith("beam.hdf5") for mode, path in opens)
    assert not [path for mode, path in opens[:first_ps] if mode != "w" and "/beam_m/" in path]   # neither beam nor svd files
    assert not [path for mode, path in opens if mode != "w" and path.endswith("beam.hdf5")]

    t, bt1 = pm.telescope, pm.beamtransfer
    one, two = str(tmp_path / "one"), str(tmp_path / "two")

    def load(rel, name):
        out = []
        for root in (one, two):
            with storage.File(root + "/" + rel, "r") as f:
                out.append(np.array(f[name][:]))
        return out

    for mi in range(t.mmax + 1):
        f1 = bt1._mfile(mi)
        with storage.File(f1, "r") as fa, storage.File(f1.replace(one, two, 1), "r") as fb:
            assert np.array_equal(fa["beam_m"][:], fb["beam_m"][:])
    rel = os.path.relpath(bt1.directory, one)
    a, b = load(rel + "/svdspectrum.hdf5", "singularvalues")
    assert a.shape == b.shape and np.abs(a - b).max() <= 1e-12 * a.max()
    for name in ("kl", "dk"):
        evd = os.path.relpath(pm.kltransforms[name].evdir, one)
        a, b = load(evd + "/evals.hdf5", "evals")
        assert a.shape == b.shape and np.abs(a - b).max() <= 1e-9 * np.abs(a).max()
    psd = os.path.relpath(pm.psestimators["ps"].psdir, one)
    a, b = load(psd + "/fisher.hdf5", "fisher")
    assert np.abs(a).max() > 0 and np.abs(a - b).max() <= 1e-8 * np.abs(a).max()


def test_deferred_host_copies(tmp_path, monkeypatch):
    """`Context.defer_host` / `storage.Deferred`: the copy thread copies what the tensor held when it was handed over (an
    event behind its producer), on a stream of its own, while the compute stream goes on with other tensors; the files
    written from such copies are the ones the inline path writes; the device block is let go after the copy."""
    import torch
    from driftscan_amd import device, storage

    device.reset_context()
    ctx = device.get_context()
    monkeypatch.setenv("DRIFTMI_IO_THREADS", "3")
    monkeypatch.delenv("DRIFTMI_STORAGE", raising=False)
    g = torch.Generator(device="cuda").manual_seed(3)
    src, want = [], []
    for i in range(6):
        t = torch.randn(3, 1 << 20, generator=g, device="cuda", dtype=torch.float64)
        t = torch.complex(t, t.flip(0))
        t.mul_(float(i + 1))               # the producer: the event must sit behind it
        src.append(t)
        want.append(None)

    def write(i, a, b):
        with storage.File(str(tmp_path / ("d%d.hdf5" % i)), "w") as f:
            f.create_dataset("a", data=a, **storage.compression_kwargs((1, 1 << 16)))
            f.create_dataset("b", data=b)

    for i, t in enumerate(src):
        ev = ctx.record_event()
        storage.submit(write, i, ctx.defer_host(t[:2], ev), ctx.defer_host(t[2], ev))
        # the compute stream moves on at once: work on OTHER memory while the copies run
        _ = (torch.ones(1 << 22, device="cuda") * 2).sum()
    before = torch.cuda.memory_allocated()
    want = [t.cpu().numpy() for t in src]
    del src, t
    storage.flush()
    torch.cuda.synchronize()
    assert torch.cuda.memory_allocated() <= before - 6 * 3 * (1 << 20) * 16 + (1 << 20)   # the deferred views held the blocks
    for i in range(6):
        with storage.File(str(tmp_path / ("d%d.hdf5" % i)), "r") as f:
            assert np.array_equal(f["a"][...], want[i][:2]) and np.array_equal(f["b"][...], want[i][2])


def test_bench_job_mode_two_ranks_on_one_gpu():
    """`bench.py --workload configs2 --job --gpus 2` — the north-star job on REAL ranks (two processes, gloo between them,
    both on this GPU, toy telescope): every rank takes its `_my_ms()` range through ProductManager.generate(), the line
    carries per-rank seconds, the seconds inside collectives and the number of ranks the backend saw; the ranges
    partition 0..mmax.  (What `bench.py --gpus N` attaches as `north_star.job` for N > 1.)"""
    import json
    import subprocess
    import sys

    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK")}
    res = subprocess.run([sys.executable, os.path.join(root, "bench.py"), "--workload", "configs2", "--job", "--gpus", "2",
                          "--one-gpu", "--backend", "gloo", "--share-mmax", "1"], env=env, stdout=subprocess.PIPE,
                         stderr=subprocess.PIPE, timeout=600)
    assert res.returncode == 0, res.stderr.decode()[-2000:]
    line = json.loads(res.stdout.decode().strip().splitlines()[-1])
    assert line["n_gpus"] == 2 and line["ranks_seen_by_gloo"] == 2 and line["config"]["ranks"] == 2
    per = line["per_rank"]
    assert [p["rank"] for p in per] == [0, 1]
    assert per[0]["m_lo"] == 0 and per[1]["m_lo"] == per[0]["m_hi"] + 1 and per[1]["m_hi"] == line["config"]["mmax"]
    assert sum(p["m_blocks"] for p in per) == line["config"]["mmax"] + 1
    assert all(p["seconds"] > 0 and p["collective_s"] >= 0 and p["stage_s"]["svd"] > 0 for p in per)
    assert line["job_s"] >= max(p["compute_s"] for p in per) and line["value"] > 0


def test_several_btgen_ranges_with_files(tmp_path):
    """`beam_chunk_gb` small enough for several BT-gen ranges per rank, product files on: the beam blocks of a range reach
    the writer queue as views of ONE allocation; `generate` waits for their host copies before it allocates the next
    range (`storage.wait_copies`) — at most two ranges' worth of blocks are ever alive, and every file equals the one of
    the single-range run."""
    import torch
    import yaml

    from driftscan_amd import device, manager, storage

    def conf(outdir, chunk):
        return dict(
            config=dict(beamtransfers=True, kltransform=True, psfisher=False, output_directory=str(outdir), truncate=False,
                        beam_chunk_gb=chunk),
            telescope=dict(type="PolarisedCylinder", num_freq=3, freq_start=400.0, freq_end=430.0, freq_mode="edge",
                           num_cylinders=2, cylinder_width=2.0, num_feeds=3, feed_spacing=0.4, tsys=1.0, sht_iter=3),
            kltransform=[dict(type="KLTransform", name="kl", use_foregrounds=True, threshold=0.0)],
        )

    pms = {}
    for name, chunk in (("one", 96.0), ("many", 1e-9)):     # 1e-9 GB: one m-block per range
        device.reset_context()
        c = tmp_path / (name + ".yaml")
        c.write_text(yaml.dump(conf(tmp_path / name, chunk)))
        pm = manager.ProductManager.from_config(str(c))
        torch.cuda.reset_peak_memory_stats()
        pm.generate()
        storage.flush()
        pms[name] = pm
    t, bt = pms["one"].telescope, pms["one"].beamtransfer
    one, many = str(tmp_path / "one"), str(tmp_path / "many")
    for mi in range(t.mmax + 1):
        f1 = bt._mfile(mi)
        with storage.File(f1, "r") as fa, storage.File(f1.replace(one, many, 1), "r") as fb:
            assert np.array_equal(fa["beam_m"][:], fb["beam_m"][:])
            assert int(fb.attrs["sht_iter"]) == 3 and not bool(fb.attrs["sht_ring_weights"])
        s1 = bt._svdfile(mi)
        with storage.File(s1, "r") as fa, storage.File(s1.replace(one, many, 1), "r") as fb:
            a, b = fa["singularvalues"][:], fb["singularvalues"][:]
            assert np.abs(a - b).max() <= 1e-12 * max(a.max(), 1e-300)
    import yaml as _y

    dump = _y.safe_load(open(os.path.join(many, "configdump.yaml")))
    assert dump["driftscan_amd"]["sht_iter"] == 3
