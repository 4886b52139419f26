#!/usr/bin/env python3
"""Generate the committed golden vectors under tests/golden/ by running the
UNMODIFIED reference (/root/reference) through the stand-in packages in
oracle/refstubs.  Runs only in the build container (the reference tree and
oracle/_ref must be present):

    make -C oracle ref && python oracle/gen_golden.py

The fixtures are data only: seeded inputs and the reference's outputs.
TEST INFRASTRUCTURE — never imported by the product.
"""
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(HERE))

from oracle import refimport  # noqa: E402

OUT = os.path.join(os.path.dirname(HERE), "tests", "golden")


# ----------------------------------------------------------------------------
# helpers
# ----------------------------------------------------------------------------
class FakeTelescope(object):
    """Exposes exactly the attributes BeamTransfer/KLTransform read
     on top of injected arrays."""

    def __init__(self, nfreq, nbase, npol, lmax, mmax, npower, tsys_flat=1.0):
        self.nfreq = nfreq
        self.nbase = nbase
        self.npairs = nbase
        self.num_pol_sky = npol
        self._npol_sky_ = npol
        self.lmax = lmax
        self.mmax = mmax
        self.included_freq = np.arange(nfreq)
        self.included_baseline = np.arange(nbase)
        self.included_pol = np.arange(npol)
        self.frequencies = np.linspace(400.0, 450.0, nfreq)
        self.baselines = np.zeros((nbase, 2))
        self.tsys_flat = tsys_flat
        self._npower = npower  # (nfreq, nbase)

    def noisepower(self, bl_indices, f_indices, ndays=None):
        bl, fi = np.broadcast_arrays(bl_indices, f_indices)
        return self._npower[fi, bl]


def synth_beam_m(rng, F, B, P, L, m, polrank=None, polscale=0.05):
    """Random m-block (F,2,B,P,L), zero for l < m; polarised part low rank."""
    T = 2 * B
    bm = np.zeros((F, T, P, L), dtype=np.complex128)
    Lm = L - m
    for f in range(F):
        # smoothly decaying column weights so that the spectrum spans decades
        colw = np.exp(-np.arange(Lm) / max(Lm / 6.0, 1.0))
        a = (rng.standard_normal((T, Lm)) + 1j * rng.standard_normal((T, Lm))) * colw
        bm[f, :, 0, m:] = a
        if P > 1:
            r = polrank if polrank is not None else max(T // 4, 1)
            left = rng.standard_normal((T, r)) + 1j * rng.standard_normal((T, r))
            for p in range(1, P):
                right = rng.standard_normal((r, Lm)) + 1j * rng.standard_normal((r, Lm))
                bm[f, :, p, m:] = polscale * (left @ right) / np.sqrt(r)
    return bm.reshape(F, 2, B, P, L)


def analytic_cl(frequencies, L, P, kind, fg_amp=10.0):
    """The committed analytic C_l(nu,nu') model; shape (P,P,L,F,F)."""
    nu = np.asarray(frequencies, dtype=np.float64)
    F = nu.size
    ell = np.arange(L, dtype=np.float64)
    cv = np.zeros((P, P, L, F, F))
    dnu = nu[:, None] - nu[None, :]
    if kind == "signal":
        corr = np.exp(-0.5 * (dnu / 2.0) ** 2)
        cv[0, 0] = 1e-7 * (1.0 / (ell + 1.0))[:, None, None] * corr[None]
    else:
        lognu = np.log(nu[:, None] / nu[None, :])
        corr = (nu[:, None] * nu[None, :] / 408.0**2) ** -2.8 * np.exp(-0.5 * lognu**2 / 4.0**2)
        amp = fg_amp * ((ell + 1.0) / 100.0) ** -2.4
        cv[0, 0] = amp[:, None, None] * corr[None]
        if P >= 3:
            corrp = (nu[:, None] * nu[None, :] / 408.0**2) ** -2.8 * np.exp(-0.5 * lognu**2 / 0.5**2)
            cv[1, 1] = 0.05 * amp[:, None, None] * corrp[None]
            cv[2, 2] = 0.05 * amp[:, None, None] * corrp[None]
    return cv


def write_beam_file(ref, bt, mi, beam_m):
    import h5py  # the in-memory stand-in

    f = h5py.File(bt._mfile(mi), "w")
    f.create_dataset("beam_m", data=beam_m[..., mi:])
    f.close()


def read_svd_file(bt, mi):
    import h5py

    f = h5py.File(bt._svdfile(mi), "r")
    out = {k: f[k][:] for k in ("beam_svd", "invbeam_svd", "beam_ut", "singularvalues")}
    f.close()
    return out


# ----------------------------------------------------------------------------
# fixtures
# ----------------------------------------------------------------------------
def gen_matrix_ops(ref):
    rng = np.random.default_rng(1001)
    bt = ref["beamtransfer"]
    out = {}
    cases = {
        "rand_wide": rng.standard_normal((12, 30)) + 1j * rng.standard_normal((12, 30)),
        "rand_tall": rng.standard_normal((20, 7)) + 1j * rng.standard_normal((20, 7)),
        "lowrank": (rng.standard_normal((16, 4)) + 1j * rng.standard_normal((16, 4)))
        @ (rng.standard_normal((4, 24)) + 1j * rng.standard_normal((4, 24))),
        "empty": np.zeros((0, 9), dtype=np.complex128),
    }
    for name, A in cases.items():
        for rtol in (1e-10, 1e-4, 0.0):
            tag = "%s_r%g" % (name, rtol)
            img, s = bt.matrix_image(A.copy(), rtol=rtol)
            nul, s2 = bt.matrix_nullspace(A.copy(), rtol=rtol)
            out["A_" + name] = A
            out["img_" + tag] = img
            out["imgs_" + tag] = s
            out["nul_" + tag] = nul
            out["nuls_" + tag] = s2
    np.savez_compressed(os.path.join(OUT, "matrix_ops.npz"), **out)
    print("matrix_ops.npz", len(out))


def gen_svd_kl(ref, tag, F, B, P, lmax, mlist, polsvcut, seed, fg_threshold, threshold, fg_amp):
    """SVD chain -> covariance projections -> KL / DoubleKL, all by the reference."""
    btmod, klmod, dkmod = ref["beamtransfer"], ref["kltransform"], ref["doublekl"]
    rng = np.random.default_rng(seed)
    L = lmax + 1
    mmax = lmax
    # baseline-dependent white noise, the same for +m and -m
    redundancy = rng.integers(1, 6, size=B).astype(np.float64)
    npower = (2.5e-7 * (1.0 + 0.1 * np.arange(F))[:, None] / redundancy[None, :]).astype(np.float64)
    tel = FakeTelescope(F, B, P, lmax, mmax, npower, tsys_flat=1.0)

    bt = btmod.BeamTransfer("/mem/%s/bt" % tag, telescope=tel)
    bt.polsvcut = polsvcut
    bt.svcut = 1e-6

    cv_sg = analytic_cl(tel.frequencies, L, P, "signal")
    cv_fg = analytic_cl(tel.frequencies, L, P, "foreground", fg_amp=fg_amp)

    out = dict(
        F=F, B=B, P=P, lmax=lmax, mlist=np.array(mlist), polsvcut=polsvcut, svcut=bt.svcut,
        npower=npower, frequencies=tel.frequencies, cv_sg=cv_sg, cv_fg=cv_fg,
        fg_threshold=fg_threshold, threshold=threshold, tsys_flat=tel.tsys_flat, fg_amp=fg_amp,
    )

    kl = klmod.KLTransform(bt, subdir="kl")
    kl._cvsg, kl._cvfg = cv_sg, cv_fg
    kl.threshold = threshold
    klnf = klmod.KLTransform(bt, subdir="klnf")
    klnf._cvsg, klnf._cvfg = cv_sg, cv_fg
    klnf.use_foregrounds = False
    klnf.threshold = threshold
    dk = dkmod.DoubleKL(bt, subdir="dk")
    dk._cvsg, dk._cvfg = cv_sg, cv_fg
    dk.foreground_threshold = fg_threshold
    dk.threshold = threshold

    for mi in mlist:
        beam = synth_beam_m(rng, F, B, P, L, mi)
        write_beam_file(ref, bt, mi, beam)
        bt._generate_svdfile_m(mi)
        svd = read_svd_file(bt, mi)
        svnum, svbounds = bt._svd_num(mi)
        pre = "m%d_" % mi
        out[pre + "beam_m"] = beam
        for k, v in svd.items():
            out[pre + k] = v
        out[pre + "svnum"] = svnum
        out[pre + "svbounds"] = svbounds
        out[pre + "proj_sg"] = bt.project_matrix_sky_to_svd(mi, cv_sg)
        out[pre + "proj_fg"] = bt.project_matrix_sky_to_svd(mi, cv_fg)
        out[pre + "proj_sg_temponly"] = bt.project_matrix_sky_to_svd(mi, cv_sg, temponly=True)
        npw = np.concatenate([npower, npower], axis=1)
        out[pre + "proj_noise"] = bt.project_matrix_diagonal_telescope_to_svd(mi, npw)

        for name, obj in (("kl", kl), ("klnf", klnf)):
            cs, cn = obj.sn_covariance(mi)
            out[pre + name + "_cs"] = cs
            out[pre + name + "_cn"] = cn
            evals, evecs, inv, extra = obj._transform_m(mi)
            out[pre + name + "_evals"] = evals
            out[pre + name + "_evecs"] = evecs
            out[pre + name + "_ac"] = extra["ac"]
            i_ev = np.searchsorted(evals, obj.threshold)
            out[pre + name + "_nkept"] = evals.size - i_ev

        dk.use_thermal = True
        evals, evecs, inv, extra = dk._transform_m(mi)
        out[pre + "dk_evals"] = evals
        out[pre + "dk_evecs"] = evecs
        out[pre + "dk_f_evals"] = extra["f_evals"]
        out[pre + "dk_ac"] = extra["ac"]
        # the two covariance pairs DoubleKL used
        dk.use_thermal = False
        cs0, cn0 = dk.sn_covariance(mi)
        dk.use_thermal = True
        cs1, cn1 = dk.sn_covariance(mi)
        out[pre + "dk_cs_nothermal"] = cs0
        out[pre + "dk_cn_nothermal"] = cn0
        out[pre + "dk_cs_thermal"] = cs1
        out[pre + "dk_cn_thermal"] = cn1
        print(tag, "m", mi, "ndof", svbounds[-1], "kl kept", out[pre + "kl_nkept"],
              "dk modes", evals.size, "of", extra["f_evals"].size)

    np.savez_compressed(os.path.join(OUT, "svdkl_%s.npz" % tag), **out)
    print("svdkl_%s.npz" % tag)


def gen_eigh_gen(ref):
    kl = ref["kltransform"]
    rng = np.random.default_rng(77)
    out = {}
    n = 24
    X = rng.standard_normal((n, n)) + 1j * rng.standard_normal((n, n))
    A = X @ X.T.conj()
    Y = rng.standard_normal((n, n)) + 1j * rng.standard_normal((n, n))
    B = Y @ Y.T.conj() + 0.1 * np.eye(n)
    ev, evec, ac = kl.eigh_gen(A.copy(), B.copy())
    out.update(pd_A=A, pd_B=B, pd_evals=ev, pd_evecs=evec, pd_ac=ac)
    # B numerically not positive definite -> rescue branch (kltransform.py:101-111)
    Yr = rng.standard_normal((n, n - 3)) + 1j * rng.standard_normal((n, n - 3))
    Bn = Yr @ Yr.T.conj()
    Bn -= 1e-9 * np.eye(n)
    ev, evec, ac = kl.eigh_gen(A.copy(), Bn.copy())
    out.update(npd_A=A, npd_B=Bn, npd_evals=ev, npd_evecs=evec, npd_ac=ac)
    # all-zero A shortcut (kltransform.py:81-85)
    Z = np.zeros((5, 5), dtype=np.complex128)
    ev, evec, ac = kl.eigh_gen(Z.copy(), B[:5, :5].copy())
    out.update(zero_A=Z, zero_B=B[:5, :5], zero_evals=ev, zero_evecs=evec, zero_ac=ac)
    np.savez_compressed(os.path.join(OUT, "eigh_gen.npz"), **out)
    print("eigh_gen.npz ac(npd) =", out["npd_ac"])


def gen_geometry(ref):
    cyl = ref["cylinder"]
    tmod = ref["telescope"]
    out = {}
    cfgs = {
        "testparams": ("pol", dict(num_freq=8, freq_start=400.0, freq_end=450.0, freq_mode="edge",
                                    num_cylinders=2, cylinder_width=5.0, num_feeds=5, feed_spacing=0.5, tsys=1.0)),
        "cfg2": ("unpol", dict(num_freq=16, freq_start=400.0, freq_end=450.0, freq_mode="edge",
                               num_cylinders=2, cylinder_width=5.0, num_feeds=16, feed_spacing=0.4, tsys=1.0,
                               force_lmax=128, force_mmax=128)),
        "cfg3": ("pol", dict(num_freq=64, freq_start=400.0, freq_end=500.0, freq_mode="edge",
                             num_cylinders=4, cylinder_width=12.0, num_feeds=16, feed_spacing=0.4, tsys=1.0,
                             force_lmax=512, force_mmax=512)),
        # BASELINE configs[4]: CHIME-like stress case; the 512 x 512 feed tables are left out
        "cfg5": ("pol", dict(num_freq=256, freq_start=400.0, freq_end=800.0, freq_mode="edge",
                             num_cylinders=4, cylinder_width=14.5, num_feeds=64, feed_spacing=0.3, tsys=1.0,
                             force_lmax=1024, force_mmax=1024)),
        "skip": ("pol", dict(num_freq=8, freq_start=400.0, freq_end=450.0, freq_mode="edge",
                             num_cylinders=2, cylinder_width=5.0, num_feeds=5, feed_spacing=0.5, tsys=1.0,
                             skip_freq=[0, 3, 4], skip_baselines=[17, 18, 25])),
        "noincyl": ("unpol", dict(num_freq=4, freq_start=600.0, freq_end=700.0, freq_mode="centre",
                                  num_cylinders=3, cylinder_width=10.0, num_feeds=4, feed_spacing=0.6,
                                  in_cylinder=False, tsys=50.0)),
    }
    for name, (kind, cfg) in cfgs.items():
        klass = cyl.PolarisedCylinderTelescope if kind == "pol" else cyl.UnpolarisedCylinderTelescope
        t = klass.from_config(cfg)
        nat_l, nat_m = tmod.max_lm(t.baselines, t.wavelengths.min(), t.u_width, t.v_width)
        bl = np.arange(t.nbase)
        out[name + "_kind"] = kind
        out[name + "_cfg_keys"] = np.array(sorted(cfg.keys()))
        out[name + "_cfg_vals"] = np.array([repr(cfg[k]) for k in sorted(cfg.keys())])
        out[name + "_feedpositions"] = t.feedpositions
        out[name + "_beamclass"] = t.beamclass
        out[name + "_uniquepairs"] = t.uniquepairs
        out[name + "_baselines"] = t.baselines
        out[name + "_redundancy"] = t.redundancy
        if name != "cfg5":
            out[name + "_feedmap"] = t.feedmap
            out[name + "_feedmask"] = t.feedmask
            out[name + "_feedconj"] = t.feedconj
        out[name + "_frequencies"] = t.frequencies
        out[name + "_wavelengths"] = t.wavelengths
        out[name + "_lmax"] = t.lmax
        out[name + "_mmax"] = t.mmax
        out[name + "_natural_lmax"] = int(nat_l.max())
        out[name + "_natural_mmax"] = int(nat_m.max())
        out[name + "_noisepower"] = np.array(
            [np.asarray(t.noisepower(bl, fi)).reshape(-1) for fi in range(t.nfreq)]
        )
        out[name + "_included_freq"] = t.included_freq
        out[name + "_included_baseline"] = t.included_baseline
        out[name + "_included_pol"] = t.included_pol
        out[name + "_zenith"] = t.zenith
        # per-(baseline, freq) band limits used by transfer_matrices (telescope.py:792-802)
        bb, ff = [a.ravel() for a in np.meshgrid(bl, np.arange(t.nfreq), indexing="ij")]
        lm, mm = np.ceil(t.l_boost * np.array(tmod.max_lm(
            t.baselines[bb], t.wavelengths[ff], t.u_width, t.v_width
        ))).astype(np.int64)
        out[name + "_lmax_bf"] = lm.reshape(t.nbase, t.nfreq)
        out[name + "_mmax_bf"] = mm.reshape(t.nbase, t.nfreq)
        print(name, "nfeed", t.nfeed, "nbase", t.nbase, "lmax", t.lmax, "mmax", t.mmax,
              "natural", int(nat_l.max()), int(nat_m.max()))
    np.savez_compressed(os.path.join(OUT, "geometry.npz"), **out)
    print("geometry.npz")


def gen_pixel_kernels(ref):
    """The reference's compiled Cython kernels and the cylinder beam model on a
    seeded set of sky positions."""
    ft = ref["fast_tools"]
    cylbeam = ref["cylbeam"]
    vis = ref["visibility"]
    rng = np.random.default_rng(5)
    n = 3000
    theta = np.arccos(rng.uniform(-1, 1, n))
    phi = rng.uniform(0, 2 * np.pi, n)
    angpos = np.stack([theta, phi], axis=-1)
    zenith = np.array([np.pi / 2.0 - np.radians(45.0), 0.0])
    uv = np.array([7.3, -11.9])
    out = dict(angpos=angpos, zenith=zenith, uv=uv)
    fr = ft.fringe(angpos, zenith, uv)
    out["fringe"] = fr
    hz = vis.horizon(angpos, zenith)
    out["horizon"] = hz
    st = rng.uniform(-1, 1, 500)
    out["exptan_in"] = st
    out["exptan_fwhm"] = 1.3
    out["exptan"] = ft.beam_exptan(st, 1.3)
    width, fe, fh = 5.0 / 0.7, 2.0 * np.pi / 3.0 * 0.7, 2.0 * np.pi / 3.0 * 1.0
    out["cyl_width"], out["cyl_fwhm_e"], out["cyl_fwhm_h"] = width, fe, fh
    out["beam_amp"] = cylbeam.beam_amp(angpos, zenith, width, fh, fh)
    bx = cylbeam.beam_x(angpos, zenith, width, fe, fh)
    by = cylbeam.beam_y(angpos, zenith, width, fe, fh)
    out["beam_x"], out["beam_y"] = bx, by
    pat = cylbeam.fraunhofer_cylinder(lambda t: ft.beam_exptan(t, fh), width)
    out["fraunhofer_x"], out["fraunhofer_y"], out["fraunhofer_y2"] = pat.x, pat.y, pat.y2
    hzf = hz.astype(np.float64)
    out["pol_real_xy"] = ft._construct_pol_real(bx, by, fr, hzf)
    out["pol_real_xx"] = ft._construct_pol_real(bx, bx, fr, hzf)
    bxc = bx * np.exp(0.3j)
    byc = by * np.exp(-0.7j)
    out["beam_xc"], out["beam_yc"] = bxc, byc
    out["pol_complex_xy"] = ft._construct_pol_complex(bxc, byc, fr, hzf)
    np.savez_compressed(os.path.join(OUT, "pixel_kernels.npz"), **out)
    print("pixel_kernels.npz")


def _telescope_cases(ref):
    """(name, constructor) of the reference's other telescope classes (drift/telescope/*.py, examples/disharray)."""
    import importlib

    T = {n: importlib.import_module("drift.telescope." + n) for n in ["gmrt", "restrictedcylinder", "exotic_cylinder"]}
    exdir = os.path.join(refimport.REFROOT, "examples", "disharray")
    if exdir not in sys.path:
        sys.path.insert(0, exdir)
    import simplearray

    band = dict(num_freq=3, freq_start=400.0, freq_end=450.0, freq_mode="edge", tsys=1.0)
    cyl = dict(band, num_cylinders=2, num_feeds=4, cylinder_width=5.0, feed_spacing=0.5)
    return [
        ("gmrt", lambda: T["gmrt"].GmrtUnpolarised(pointing=5.0), dict(pointing=5.0)),
        ("restricted_box", lambda c: T["restrictedcylinder"].RestrictedCylinder.from_config(c), dict(cyl, beam_height=40.0)),
        ("restricted_pol_gauss", lambda c: T["restrictedcylinder"].RestrictedPolarisedCylinder.from_config(c),
         dict(cyl, beam_type="gaussian", beam_height=25.0)),
        ("restricted_extra", lambda c: T["restrictedcylinder"].RestrictedExtra.from_config(c),
         dict(cyl, extra_feeds=[-1.3, -2.9])),
        ("random", lambda c: T["exotic_cylinder"].RandomCylinder.from_config(c), dict(cyl)),
        ("gradient", lambda c: T["exotic_cylinder"].GradientCylinder.from_config(c), dict(cyl, max_spacing=6.0)),
        ("extra", lambda c: T["exotic_cylinder"].CylinderExtra.from_config(c), dict(cyl, extra_feeds=[-1.3])),
        ("perturbed", lambda c: T["exotic_cylinder"].CylinderPerturbed.from_config(c), dict(cyl, num_feeds=3)),
        ("dish_pol", lambda c: simplearray.DishArray.from_config(c), dict(freq_mode="edge", tsys=1.0)),   # the example's band (100-150 MHz, 5 channels) and 4 x 4 grid are class attributes
    ]


def gen_telescopes(ref):
    """Geometry, host beams and visibility-response maps (before the spherical-harmonic transform, which is
    unpinned) of the reference's non-cylinder / modified-cylinder telescope classes.  ``_init_trans`` needs
    healpy's pix2ang through cora: the pixel centres are supplied by the oracle's restatement instead."""
    from oracle import btgen as obt

    vis = ref["visibility"]
    out = {}
    nside = 16
    ap = obt.ang_positions(nside)
    names = []
    for name, make, cfg in _telescope_cases(ref):
        t = make() if name == "gmrt" else make(cfg)
        names.append(name)
        out[name + "_cfg_keys"] = np.array(sorted(cfg.keys()))
        out[name + "_cfg_vals"] = np.array([repr(cfg[k]) for k in sorted(cfg.keys())])
        for attr in ("feedpositions", "beamclass", "uniquepairs", "baselines", "redundancy", "feedmap", "feedmask",
                     "feedconj", "frequencies", "zenith"):
            out[name + "_" + attr] = np.asarray(getattr(t, attr))
        out[name + "_lmax"], out[name + "_mmax"] = t.lmax, t.mmax
        out[name + "_npol"] = t.num_pol_sky
        bl = np.arange(t.nbase)
        out[name + "_noisepower"] = np.array([np.asarray(t.noisepower(bl, fi)).reshape(-1) for fi in range(t.nfreq)])
        t._nside, t._angpos = nside, ap
        t._horizon = vis.horizon(ap, t.zenith)
        classes = np.unique(t.beamclass)
        feeds = np.array([int(np.nonzero(t.beamclass == c)[0][0]) for c in classes])
        out[name + "_beam_feeds"] = feeds
        out[name + "_beams"] = np.array([t.beam(int(fd), 1) for fd in feeds])
        rng = np.random.default_rng(11)
        bsel = np.sort(rng.choice(t.nbase, size=min(6, t.nbase), replace=False))
        out[name + "_map_bl"] = bsel
        out[name + "_maps"] = np.array([t._beam_map_single(int(b), 1) for b in bsel])
        print(name, type(t).__name__, "nfeed", t.nfeed, "nbase", t.nbase, "lmax", t.lmax, "npol", t.num_pol_sky,
              "maps", out[name + "_maps"].shape)
    out["names"] = np.array(names)
    out["nside"] = nside
    np.savez_compressed(os.path.join(OUT, "telescopes.npz"), **out)
    print("telescopes.npz")


FGT_UNPOL, FGT_POL, THR_POL = 1.0, 1e-2, 1e-3


def gen_projections(ref):
    """A17 operators that do not depend on the SVD basis, by the reference: invbeam_m,
    telescope->sky (map-making), dirty back-projection, sky covariance -> telescope, sky -> telescope."""
    btmod = ref["beamtransfer"]
    # The reference calls scipy.linalg.pinv(..., rcond=1e-6) (blockla.py:136 via beamtransfer.py:344);
    # the scipy in this image (>= 1.14) dropped that keyword in favour of its documented successor
    # `rtol` (same meaning: cutoff relative to the largest singular value).  Translate it for the
    # duration of this generator so that the unmodified reference function runs.
    import scipy.linalg as _sla

    _pinv = _sla.pinv

    def _pinv_compat(a, *args, rcond=None, **kw):
        if rcond is not None:
            kw["rtol"] = rcond
        return _pinv(a, *args, **kw)

    _sla.pinv = _pinv_compat
    out = {}
    for tag, F, B, P, lmax, mlist, seed in (("unpol", 3, 6, 1, 14, [0, 4], 3101), ("pol", 2, 5, 4, 10, [0, 3], 3102)):
        rng = np.random.default_rng(seed)
        L = lmax + 1
        redundancy = rng.integers(1, 6, size=B).astype(np.float64)
        npower = (2.5e-7 * (1.0 + 0.1 * np.arange(F))[:, None] / redundancy[None, :]).astype(np.float64)
        tel = FakeTelescope(F, B, P, lmax, lmax, npower, tsys_flat=1.0)
        bt = btmod.BeamTransfer("/mem/proj_%s/bt" % tag, telescope=tel)
        cv = analytic_cl(tel.frequencies, L, P, "signal")
        out[tag + "_dims"] = np.array([F, B, P, lmax])
        out[tag + "_mlist"] = np.array(mlist)
        out[tag + "_npower"] = npower
        out[tag + "_cv"] = cv
        for mi in mlist:
            beam = synth_beam_m(rng, F, B, P, L, mi)
            write_beam_file(ref, bt, mi, beam)
            pre = "%s_m%d_" % (tag, mi)
            vt = rng.standard_normal((F, 2 * B)) + 1j * rng.standard_normal((F, 2 * B))
            vs = rng.standard_normal((F, P, L)) + 1j * rng.standard_normal((F, P, L))
            vs[..., :mi] = 0.0
            out[pre + "beam_m"] = beam
            out[pre + "vec_tel"] = vt
            out[pre + "vec_sky"] = vs
            out[pre + "invbeam_m"] = bt.invbeam_m(mi)
            out[pre + "tel_to_sky"] = bt.project_vector_telescope_to_sky(mi, vt)
            out[pre + "backward_dirty"] = bt.project_vector_backward_dirty(mi, vt)
            out[pre + "sky_to_tel"] = bt.project_vector_sky_to_telescope(mi, vs)
            out[pre + "mat_sky_to_tel"] = bt.project_matrix_sky_to_telescope(mi, cv)
            out[pre + "mat_sky_to_tel_temponly"] = bt.project_matrix_sky_to_telescope(mi, cv, temponly=True)
    _sla.pinv = _pinv
    np.savez_compressed(os.path.join(OUT, "projections.npz"), **out)
    print("projections.npz")


def gen_psfisher(ref):
    """G8: PSExact._work_fisher_bias_m with injected band C_l arrays (the cora band generator is
    not available; the bands are inputs).  Inputs are the products of svdkl_unpol.npz."""
    btmod, klmod, psmod = ref["beamtransfer"], ref["kltransform"], ref["psestimation"]
    from oracle import psfisher as opf

    g = np.load(os.path.join(OUT, "svdkl_unpol.npz"))
    F, B, P, lmax = int(g["F"]), int(g["B"]), int(g["P"]), int(g["lmax"])
    L = lmax + 1
    tel = FakeTelescope(F, B, P, lmax, lmax, g["npower"], tsys_flat=1.0)
    tel.frequencies = g["frequencies"]
    bt = btmod.BeamTransfer("/mem/psf/bt", telescope=tel)
    bt.polsvcut, bt.svcut = float(g["polsvcut"]), float(g["svcut"])
    kl = klmod.KLTransform(bt, subdir="kl")
    kl._cvsg, kl._cvfg = g["cv_sg"], g["cv_fg"]
    kl.threshold = float(g["threshold"])
    kl.inverse = False
    l_edges = np.array([0, 4, 9, 15, L])
    clarray = opf.band_clarray(tel.frequencies, L, l_edges)
    # the h5py stand-in keeps files in memory: let the reference's `os.path.exists` checks see them
    import h5py as _h5

    class _OsShim(object):
        def __getattr__(self, name):
            return getattr(os, name)

    class _PathShim(object):
        def __getattr__(self, name):
            return getattr(os.path, name)

        @staticmethod
        def exists(path):
            return _h5.exists(path) or os.path.exists(path)

    shim = _OsShim()
    shim.path = _PathShim()
    klmod.os = shim
    psmod.os = shim
    ps = psmod.PSExact(kl, subdir="ps")
    ps.clarray = clarray
    ps.k_center = np.arange(clarray.shape[0], dtype=np.float64)  # nbands = k_center.size
    ps.bands = list(range(clarray.shape[0] + 1))  # only its length is read (cache-file pattern, psestimation.py:668)
    ps.threshold = 0.0
    out = dict(l_edges=l_edges, clarray=clarray, ps_threshold=ps.threshold)
    for mi in [int(m) for m in g["mlist"]]:
        write_beam_file(ref, bt, mi, g["m%d_beam_m" % mi])
        bt._generate_svdfile_m(mi)
        kl.transform_save(mi)
        fisher, bias = ps.fisher_bias_m(mi)
        evals, evecs = kl.modes_m(mi, threshold=ps.threshold)
        out["m%d_fisher" % mi] = fisher
        out["m%d_bias" % mi] = bias
        out["m%d_nmodes" % mi] = 0 if evals is None else evals.size
        print("psfisher m", mi, "modes", out["m%d_nmodes" % mi], "diag", np.real(np.diag(fisher)))
    klmod.os = os
    psmod.os = os
    np.savez_compressed(os.path.join(OUT, "psfisher.npz"), **out)
    print("psfisher.npz")


def gen_bt_variants(ref):
    """BeamTransferNoSVD (covariances + KL spectrum in the telescope basis) and BeamTransferFullSVD
    (one SVD of the full polarised beam per frequency), by the reference."""
    btmod, klmod = ref["beamtransfer"], ref["kltransform"]
    out = {}
    # ---- NoSVD, unpolarised
    F, B, P, lmax, seed = 3, 5, 1, 10, 4101
    rng = np.random.default_rng(seed)
    L = lmax + 1
    redundancy = rng.integers(1, 6, size=B).astype(np.float64)
    npower = (2.5e-7 * (1.0 + 0.1 * np.arange(F))[:, None] / redundancy[None, :]).astype(np.float64)
    tel = FakeTelescope(F, B, P, lmax, lmax, npower, tsys_flat=1.0)
    bt = btmod.BeamTransferNoSVD("/mem/nosvd/bt", telescope=tel)
    cv_sg = analytic_cl(tel.frequencies, L, P, "signal")
    cv_fg = analytic_cl(tel.frequencies, L, P, "foreground", fg_amp=3e-9)
    kl = klmod.KLTransform(bt, subdir="kl")
    kl._cvsg, kl._cvfg = cv_sg, cv_fg
    kl.threshold = 0.1
    out.update(nosvd_dims=np.array([F, B, P, lmax]), nosvd_npower=npower, nosvd_cv_sg=cv_sg, nosvd_cv_fg=cv_fg,
               nosvd_mlist=np.array([0, 3]))
    for mi in [0, 3]:
        beam = synth_beam_m(rng, F, B, P, L, mi)
        write_beam_file(ref, bt, mi, beam)
        cs, cn = kl.sn_covariance(mi)
        evals, evecs, inv, extra = kl._transform_m(mi)
        vec = rng.standard_normal((F, P, L)) + 1j * rng.standard_normal((F, P, L))
        vec[..., :mi] = 0
        pre = "nosvd_m%d_" % mi
        out[pre + "beam_m"] = beam
        out[pre + "cs"], out[pre + "cn"], out[pre + "evals"] = cs, cn, evals
        out[pre + "vec_sky"] = vec
        out[pre + "sky_to_svd"] = bt.project_vector_sky_to_svd(mi, vec)
        out[pre + "svd_to_sky_conj"] = bt.project_vector_svd_to_sky(mi, bt.project_vector_sky_to_svd(mi, vec), conj=True)
        out[pre + "ndof"] = bt.ndof(mi)
    # ---- FullSVD, polarised: needs every m file
    F, B, P, lmax, seed = 2, 4, 4, 8, 4102
    rng = np.random.default_rng(seed)
    L = lmax + 1
    redundancy = rng.integers(1, 6, size=B).astype(np.float64)
    npower = (2.5e-7 * (1.0 + 0.1 * np.arange(F))[:, None] / redundancy[None, :]).astype(np.float64)
    tel = FakeTelescope(F, B, P, lmax, lmax, npower, tsys_flat=1.0)
    bt = btmod.BeamTransferFullSVD("/mem/fullsvd/bt", telescope=tel)
    out.update(fullsvd_dims=np.array([F, B, P, lmax]), fullsvd_npower=npower)
    for mi in range(lmax + 1):
        beam = synth_beam_m(rng, F, B, P, L, mi, polrank=None, polscale=0.3)
        write_beam_file(ref, bt, mi, beam)
        out["fullsvd_m%d_beam_m" % mi] = beam
    bt._generate_svdfiles()
    for mi in range(lmax + 1):
        svd = read_svd_file(bt, mi)
        for k, v in svd.items():
            out["fullsvd_m%d_%s" % (mi, k)] = v
    # ---- TempSVD, polarised: one SVD of the temperature part per frequency, every polarisation carried along
    F, B, P, lmax, seed = 2, 4, 4, 8, 4103
    rng = np.random.default_rng(seed)
    L = lmax + 1
    redundancy = rng.integers(1, 6, size=B).astype(np.float64)
    npower = (2.5e-7 * (1.0 + 0.1 * np.arange(F))[:, None] / redundancy[None, :]).astype(np.float64)
    tel = FakeTelescope(F, B, P, lmax, lmax, npower, tsys_flat=1.0)
    bt = btmod.BeamTransferTempSVD("/mem/tempsvd/bt", telescope=tel)
    out.update(tempsvd_dims=np.array([F, B, P, lmax]), tempsvd_npower=npower)
    for mi in range(lmax + 1):
        beam = synth_beam_m(rng, F, B, P, L, mi, polrank=None, polscale=0.3)
        write_beam_file(ref, bt, mi, beam)
        out["tempsvd_m%d_beam_m" % mi] = beam
    bt._generate_svdfiles()
    for mi in range(lmax + 1):
        svd = read_svd_file(bt, mi)
        for k, v in svd.items():
            out["tempsvd_m%d_%s" % (mi, k)] = v
    np.savez_compressed(os.path.join(OUT, "bt_variants.npz"), **out)
    print("bt_variants.npz")


def gen_inverse(ref):
    """`inverse = True` for KLTransform and DoubleKL, and a frequency-ASYMMETRIC sky covariance through
    project_matrix_sky_to_svd.  The beams are real-valued: the reference's DoubleKL inverse
    (doublekl.py:63-67, :83-85) multiplies an un-transposed inverse into a transposed one, so it depends on
    the arbitrary phases LAPACK gives the eigenvectors; on a real pencil those phases are signs, which
    cancel, and the output is unique."""
    btmod, klmod, dkmod = ref["beamtransfer"], ref["kltransform"], ref["doublekl"]
    F, B, P, lmax, mlist = 3, 6, 1, 16, [2, 9]
    rng = np.random.default_rng(2003)
    L = lmax + 1
    redundancy = rng.integers(1, 6, size=B).astype(np.float64)
    npower = (2.5e-7 * (1.0 + 0.1 * np.arange(F))[:, None] / redundancy[None, :]).astype(np.float64)
    tel = FakeTelescope(F, B, P, lmax, lmax, npower, tsys_flat=1.0)
    bt = btmod.BeamTransfer("/mem/inverse/bt", telescope=tel)
    bt.polsvcut, bt.svcut = 1e-4, 1e-6
    cv_sg = analytic_cl(tel.frequencies, L, P, "signal")
    cv_fg = analytic_cl(tel.frequencies, L, P, "foreground", fg_amp=3e-9)
    cv_asym = cv_fg * (1.0 + 0.3 * np.arange(F)[:, None] - 0.2 * np.arange(F)[None, :])[None, None, None]
    assert not np.array_equal(cv_asym, cv_asym.swapaxes(3, 4))
    out = dict(F=F, B=B, P=P, lmax=lmax, mlist=np.array(mlist), polsvcut=bt.polsvcut, svcut=bt.svcut, npower=npower,
               frequencies=tel.frequencies, cv_sg=cv_sg, cv_fg=cv_fg, cv_asym=cv_asym, fg_threshold=FGT_UNPOL,
               threshold=0.1, tsys_flat=1.0)
    kl = klmod.KLTransform(bt, subdir="kl")
    kl._cvsg, kl._cvfg, kl.threshold, kl.inverse = cv_sg, cv_fg, 0.1, True
    dk = dkmod.DoubleKL(bt, subdir="dk")
    dk._cvsg, dk._cvfg, dk.threshold, dk.inverse = cv_sg, cv_fg, 0.1, True
    dk.foreground_threshold = FGT_UNPOL
    for mi in mlist:
        beam = synth_beam_m(rng, F, B, P, L, mi)
        beam = np.ascontiguousarray(beam.real).astype(np.complex128)
        write_beam_file(ref, bt, mi, beam)
        bt._generate_svdfile_m(mi)
        svd = read_svd_file(bt, mi)
        pre = "m%d_" % mi
        out[pre + "beam_m"] = beam
        for k, v in svd.items():
            out[pre + k] = v
        out[pre + "proj_asym"] = bt.project_matrix_sky_to_svd(mi, cv_asym)
        evals, evecs, inv, extra = kl._transform_m(mi)
        out[pre + "kl_evals"], out[pre + "kl_evecs"], out[pre + "kl_inv"] = evals, evecs, inv
        dk.use_thermal = True
        evals, evecs, inv, extra = dk._transform_m(mi)
        assert np.abs(evecs.imag).max() <= 1e-12 * np.abs(evecs).max(), "LAPACK left complex phases on a real pencil"
        out[pre + "dk_evals"], out[pre + "dk_evecs"], out[pre + "dk_inv"] = evals, evecs, inv
        out[pre + "dk_f_evals"], out[pre + "dk_ac"] = extra["f_evals"], extra["ac"]
        print("inverse m", mi, "kl modes", out[pre + "kl_evals"].size, "dk modes", evals.size, "of", extra["f_evals"].size)
    np.savez_compressed(os.path.join(OUT, "svdkl_inverse.npz"), **out)
    print("svdkl_inverse.npz")


def gen_timestream(ref):
    """drift/pipeline/timestream.py by the unmodified reference class: the m-mode transform of a synthetic timestream
    (`generate_mmodes`, :129-185), its SVD and KL projections (`generate_mmodes_svd` :215-236, `generate_mmodes_kl`
    :331-356) and the a_lm stage of the three map-makers (`_make_alm` of `mapmake_full` :239-246, `mapmake_svd` :272-279,
    `mapmake_kl` :408-425; the synthesis `hputil.sphtrans_inv_sky` behind them is cora's and is not run).  Products of all
    m = 0 .. mmax come from the reference too (SVD chain, KL transform with `inverse`)."""
    import tempfile

    import h5py  # the in-memory stand-in
    import scipy.linalg as _sla

    from drift.pipeline import timestream as tsmod

    btmod, klmod = ref["beamtransfer"], ref["kltransform"]
    _pinv = _sla.pinv

    def _pinv_compat(a, *args, rcond=None, **kw):  # scipy >= 1.14: rcond -> rtol (see gen_projections)
        if rcond is not None:
            kw["rtol"] = rcond
        return _pinv(a, *args, **kw)

    _sla.pinv = _pinv_compat

    # the h5py stand-in keeps files in memory: let the reference's `os.path.exists` checks see them
    class _OsShim(object):
        def __getattr__(self, name):
            return getattr(os, name)

    class _PathShim(object):
        def __getattr__(self, name):
            return getattr(os.path, name)

        @staticmethod
        def exists(path):
            return h5py.exists(path) or os.path.exists(path)

    shim = _OsShim()
    shim.path = _PathShim()
    klmod.os = shim
    tsmod.os = shim
    rng = np.random.default_rng(4101)
    F, B, P, lmax = 3, 5, 4, 8
    L, mmax = lmax + 1, lmax
    redundancy = rng.integers(1, 6, size=B).astype(np.float64)
    npower = (2.5e-7 * (1.0 + 0.1 * np.arange(F))[:, None] / redundancy[None, :]).astype(np.float64)
    tel = FakeTelescope(F, B, P, lmax, mmax, npower, tsys_flat=1.0)
    bt = btmod.BeamTransfer("/mem/ts/bt", telescope=tel)
    bt.polsvcut, bt.svcut = 1e-4, 1e-6
    cv_sg = analytic_cl(tel.frequencies, L, P, "signal")
    out = dict(dims=np.array([F, B, P, lmax]), npower=npower, cv_sg=cv_sg, polsvcut=bt.polsvcut, svcut=bt.svcut)
    for mi in range(mmax + 1):
        beam = synth_beam_m(rng, F, B, P, L, mi)
        write_beam_file(ref, bt, mi, beam)
        bt._generate_svdfile_m(mi)
        out["m%d_beam_m" % mi] = beam
    kl = klmod.KLTransform(bt, subdir="kl")
    kl._cvsg, kl._cvfg = cv_sg, np.zeros_like(cv_sg)
    kl.use_foregrounds = False
    kl.inverse = True
    kl.threshold = 0.0
    # the cut of the data projection: a value that no eigenvalue of any m comes near (5 %), so that the kept count is
    # not a matter of rounding
    allev = []
    for mi in range(mmax + 1):
        kl.transform_save(mi)
        ev = kl.evals_m(mi)
        if ev is not None:
            allev.append(ev)
    allev = np.sort(np.concatenate(allev))
    cand = np.sqrt(allev[:-1] * allev[1:])
    gap = allev[1:] / allev[:-1]
    mid = np.argsort(np.abs(np.arange(cand.size) - cand.size // 2))
    thr = next(float(cand[i]) for i in mid if gap[i] > 1.1 and cand[i] > 0)
    out["kl_threshold"] = thr

    class PM(object):
        beamtransfer = bt
        kltransforms = {"kl": kl}

    tsdir = tempfile.mkdtemp(prefix="refts_")
    ts = tsmod.Timestream(tsdir, PM())
    ntime = 2 * mmax + 5
    data = rng.standard_normal((F, B, ntime)) + 1j * rng.standard_normal((F, B, ntime))
    out["timestream"] = data
    for fi in range(F):
        f = h5py.File(ts._ffile(fi), "w")
        f.create_dataset("timestream", data=data[fi])
        f.attrs["ntime"] = ntime
        f.close()
    ts.generate_mmodes()
    ts.generate_mmodes_svd()
    ts.set_kltransform("kl", threshold=thr)
    ts.generate_mmodes_kl()
    alm_full = np.zeros((F, P, L, L), dtype=np.complex128)
    alm_svd, alm_kl, alm_klw = np.zeros_like(alm_full), np.zeros_like(alm_full), np.zeros_like(alm_full)
    svd_norm, nkl = np.zeros(mmax + 1), np.zeros(mmax + 1, dtype=np.int64)
    for mi in range(mmax + 1):
        mm = ts.mmode(mi)
        out["m%d_mmode" % mi] = mm
        alm_full[..., mi] = bt.project_vector_telescope_to_sky(mi, mm)
        sv = ts.mmode_svd(mi)
        svd_norm[mi] = np.linalg.norm(sv)
        alm_svd[..., mi] = bt.project_vector_svd_to_sky(mi, sv)
        klm = ts.mmode_kl(mi)
        nkl[mi] = klm.size
        if mi >= 1:  # no_m_zero
            alm_kl[..., mi] = bt.project_vector_svd_to_sky(mi, kl.project_vector_kl_to_svd(mi, klm.copy(), threshold=thr))
            kw = klm.copy()
            ev = kl.evals_m(mi, thr)
            if ev is not None:
                kw *= ev / (1.0 + ev)
            alm_klw[..., mi] = bt.project_vector_svd_to_sky(mi, kl.project_vector_kl_to_svd(mi, kw, threshold=thr))
    out.update(alm_full=alm_full, alm_svd=alm_svd, alm_kl=alm_kl, alm_kl_wiener=alm_klw, svd_norm=svd_norm, nkl=nkl)
    _sla.pinv = _pinv
    klmod.os = os
    tsmod.os = os
    np.savez_compressed(os.path.join(OUT, "timestream.npz"), **out)
    print("timestream.npz", "threshold", thr, "kl modes per m", nkl.tolist())


def main():
    os.makedirs(OUT, exist_ok=True)
    ref = refimport.load()
    if len(sys.argv) > 1 and sys.argv[1] == "inverse":
        gen_inverse(ref)
        return
    if len(sys.argv) > 1 and sys.argv[1] == "telescopes":
        gen_telescopes(ref)
        return
    if len(sys.argv) > 1 and sys.argv[1] == "variants":
        gen_bt_variants(ref)
        return
    if len(sys.argv) > 1 and sys.argv[1] == "psfisher":
        gen_psfisher(ref)
        return
    if len(sys.argv) > 1 and sys.argv[1] == "projections":
        gen_projections(ref)
        return
    if len(sys.argv) > 1 and sys.argv[1] == "geometry":
        gen_geometry(ref)
        return
    if len(sys.argv) > 1 and sys.argv[1] == "timestream":
        gen_timestream(ref)
        return
    gen_matrix_ops(ref)
    gen_eigh_gen(ref)
    gen_geometry(ref)
    gen_pixel_kernels(ref)
    gen_projections(ref)
    # moderately conditioned pencils (cond(N) ~ 1e5): 1e-10 parity is attainable by any
    # backward-stable solver
    gen_svd_kl(ref, "unpol", F=4, B=10, P=1, lmax=24, mlist=[0, 5, 20], polsvcut=1e-4,
               seed=2001, fg_threshold=FGT_UNPOL, threshold=0.1, fg_amp=3e-9)
    gen_svd_kl(ref, "pol", F=3, B=8, P=4, lmax=20, mlist=[0, 4, 17], polsvcut=1e-4,
               seed=2002, fg_threshold=FGT_POL, threshold=THR_POL, fg_amp=3e-9)
    # realistic foreground amplitude: cond(N) ~ 1e14, the generalised eigenproblem is
    # ill-conditioned and LAPACK's own answer carries an eps*cond error bar
    gen_svd_kl(ref, "unpol_harsh", F=4, B=10, P=1, lmax=24, mlist=[5], polsvcut=1e-4,
               seed=2001, fg_threshold=1.0, threshold=0.1, fg_amp=10.0)
    gen_psfisher(ref)
    gen_inverse(ref)
    gen_telescopes(ref)
    gen_timestream(ref)


if __name__ == "__main__":
    main()
