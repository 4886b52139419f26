"""Consumers of the operators: timestream simulation, the m-mode transform and the
map-makers of drift/pipeline/timestream.py on top of the GPU-backed BeamTransfer / KLTransform.
`test_against_reference_timestream` pins the m-mode transform, the SVD / KL projections of the data and the a_lm stage of
the three map-makers on outputs of the unmodified reference class (tests/golden/timestream.npz).  cora's sky <-> a_lm
transforms are not available, so the synthesis to maps and `simulate` are checked through the identities the
reference's code path implies, on a small polarised cylinder."""
import os

import numpy as np
import pytest

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def prod(tmp_path_factory):
    import yaml

    from driftscan_amd import device, manager

    device.reset_context()
    d = tmp_path_factory.mktemp("ts")
    conf = dict(config=dict(beamtransfers=True, kltransform=True, psfisher=False, output_directory=str(d / "prod"), truncate=False),
                telescope=dict(type="PolarisedCylinder", num_freq=3, freq_start=400.0, freq_end=430.0, freq_mode="edge",
                               num_cylinders=2, cylinder_width=2.0, num_feeds=3, feed_spacing=0.4, tsys=1.0),
                kltransform=[dict(type="KLTransform", name="kl", threshold=0.0, inverse=True, use_foregrounds=False)])
    cfile = str(d / "params.yaml")
    open(cfile, "w").write(yaml.dump(conf))
    pm = manager.ProductManager.from_config(cfile)
    pm.generate()
    return pm, d


def test_sky_transforms_round_trip(prod):
    """sphtrans_inv_sky then sphtrans_sky on a band-limited polarised sky: the coefficients come back to the accuracy
    of the equal-weight quadrature (l well below the pixel scale)."""
    from driftscan_amd import healpix

    rng = np.random.default_rng(3)
    nside, lmax = 32, 12
    alm = np.zeros((2, 4, lmax + 1, lmax + 1), dtype=np.complex128)
    for l in range(lmax + 1):
        for m in range(l + 1):
            alm[:, :, l, m] = rng.standard_normal((2, 4)) + (1j * rng.standard_normal((2, 4)) if m else 0)
    alm[:, 1:3, :2] = 0.0
    maps = healpix.sphtrans_inv_sky(alm, nside)
    assert maps.shape == (2, 4, 12 * nside * nside) and maps.dtype == np.float64
    back = healpix.sphtrans_sky(maps, lmax)
    assert back.shape == alm.shape
    assert np.abs(back - alm).max() < 2e-3 * np.abs(alm).max()
    assert not np.abs(np.triu(np.ones((lmax + 1, lmax + 1)), 1) * back).any()     # m > l stays empty


def test_simulate_mmodes_and_maps(prod):
    from driftscan_amd import healpix, storage, timestream

    pm, d = prod
    bt, tel = pm.beamtransfer, pm.telescope
    rng = np.random.default_rng(4)
    nside = 32
    lmax, mmax = tel.lmax, tel.mmax
    alm = np.zeros((tel.nfreq, 4, lmax + 1, lmax + 1), dtype=np.complex128)
    for l in range(lmax + 1):
        for m in range(l + 1):
            alm[:, :, l, m] = rng.standard_normal((tel.nfreq, 4)) + (1j * rng.standard_normal((tel.nfreq, 4)) if m else 0)
    alm[:, 1:3, :2] = 0.0
    skyfile = str(d / "sky.hdf5")
    with storage.File(skyfile, "w") as f:
        f.create_dataset("map", data=healpix.sphtrans_inv_sky(alm, nside))
    ts = timestream.simulate(pm, str(d / "ts"), maps=[skyfile], ndays=0)
    assert ts.ntime == 2 * mmax + 1
    v = ts.timestream_f(1)
    assert v.shape == (tel.npairs, ts.ntime) and v.dtype == np.complex128
    with storage.File(ts._ffile(0), "r") as f:
        assert sorted(f.keys()) == ["baselines", "feedconj", "feedmap", "feedmask", "phi", "timestream", "uniquepairs"]
    # m-modes: the FFT of the timestream gives back what the beam made of the sky's a_lm, +m and conj(-m) packed
    ts.generate_mmodes()
    assert os.path.exists(ts.output_directory + "/mmodes/COMPLETED_M")
    alm_in = healpix.sphtrans_sky(healpix.sphtrans_inv_sky(alm, nside), lmax)      # what simulate() transformed
    for mi in (0, 1, mmax // 2, mmax):
        mm = ts.mmode(mi)
        assert mm.shape == (tel.nfreq, 2, tel.npairs)
        want = bt.project_vector_sky_to_telescope(mi, np.ascontiguousarray(alm_in[..., mi])).reshape(tel.nfreq, 2, tel.npairs)
        if mi == 0:
            want[:, 1] = 0.0
        assert np.abs(mm - want).max() < 1e-10 * max(np.abs(want).max(), 1e-300), mi
    # SVD and KL projections of the data, then the three map-makers
    ts.generate_mmodes_svd()
    sv = ts.mmode_svd(2)
    assert sv.shape == (int(bt.ndof(2)),)
    assert np.abs(sv - bt.project_vector_telescope_to_svd(2, ts.mmode(2).reshape(tel.nfreq, -1))).max() == 0.0
    ts.set_kltransform("kl")
    ts.generate_mmodes_kl()
    ts.collect_mmodes_kl()
    kl = pm.kltransforms["kl"]
    klm = ts.mmode_kl(2)
    assert np.abs(klm - kl.project_vector_svd_to_kl(2, sv, threshold=ts.klthreshold)).max() == 0.0
    with storage.File(ts.output_directory + ("/klmodes_kl_%f.hdf5" % ts.klthreshold), "r") as f:
        assert f["evals"].shape == (mmax + 1, bt.ndofmax)
    for name, make in (("full", lambda: ts.mapmake_full(nside, "map_full.hdf5")),
                       ("svd", lambda: ts.mapmake_svd(nside, "map_svd.hdf5")),
                       ("kl", lambda: ts.mapmake_kl(nside, "map_kl.hdf5"))):
        make()
        with storage.File(ts.output_directory + "/map_%s.hdf5" % name, "r") as f:
            mp = f["map"][:]
        assert mp.shape == (tel.nfreq, 4, 12 * nside * nside) and np.isfinite(mp).all() and np.abs(mp).max() > 0
    # the full map is the synthesis of B^+ (B a) per m
    a2 = np.zeros_like(alm)
    for mi in range(mmax + 1):
        a2[..., mi] = bt.project_vector_telescope_to_sky(mi, ts.mmode(mi))
    with storage.File(ts.output_directory + "/map_full.hdf5", "r") as f:
        assert np.abs(f["map"][:] - healpix.sphtrans_inv_sky(a2, nside)).max() < 1e-9 * np.abs(a2).max()
    # noise realisation: reproducible with a seed, and the object survives a save / load cycle
    n1 = timestream.simulate(pm, str(d / "ts_n1"), ndays=10, seed=5).timestream_f(0)
    n2 = timestream.simulate(pm, str(d / "ts_n2"), ndays=10, seed=5).timestream_f(0)
    assert np.array_equal(n1, n2) and np.abs(n1).max() > 0
    again = timestream.Timestream.load(ts.directory)
    assert again.ntime == ts.ntime and np.array_equal(again.mmode(1), ts.mmode(1))


class _FixtureTelescope(object):
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


def test_against_reference_timestream(golden_dir, tmp_path):
    """`Timestream` against outputs of the UNMODIFIED reference class (drift/pipeline/timestream.py driven through
    oracle/refstubs by oracle/gen_golden.py -> tests/golden/timestream.npz): the m-mode transform of the same synthetic
    timestream (`generate_mmodes`), the SVD and KL projections of the data and the a_lm stage of `mapmake_full`,
    `mapmake_svd` and `mapmake_kl` (with and without the Wiener weights) on the same beam blocks.  Vectors in the SVD /
    KL bases are gauge dependent and are compared through their norms, counts and their images on the sky."""
    from driftscan_amd import beamtransfer, device, kltransform, storage, timestream

    device.reset_context()
    g = np.load(os.path.join(golden_dir, "timestream.npz"))
    F, B, P, lmax = (int(x) for x in g["dims"])
    tel = _FixtureTelescope(F, B, P, lmax, g["npower"])
    bt = beamtransfer.BeamTransfer(str(tmp_path / "bt"), telescope=tel)
    bt.polsvcut, bt.svcut = float(g["polsvcut"]), float(g["svcut"])
    bt._generate_dirs()
    for mi in range(lmax + 1):
        with storage.File(bt._mfile(mi), "w") as f:
            f.create_dataset("beam_m", data=g["m%d_beam_m" % mi][..., mi:])
    bt._generate_svdfiles(regen=True)
    kl = kltransform.KLTransform.from_config(dict(threshold=0.0, inverse=True, use_foregrounds=False), bt, subdir="kl")
    kl._cvsg, kl._cvfg = g["cv_sg"], np.zeros_like(g["cv_sg"])
    kl.generate(regen=True)

    class PM(object):
        beamtransfer = bt
        kltransforms = {"kl": kl}

    ts = timestream.Timestream(str(tmp_path / "ts"), PM())
    data = g["timestream"]
    for fi in range(F):
        os.makedirs(ts._fdir(fi), exist_ok=True)
        with storage.File(ts._ffile(fi), "w") as f:
            f.create_dataset("timestream", data=data[fi])
            f.attrs["ntime"] = data.shape[-1]
    assert ts.ntime == data.shape[-1]
    ts.generate_mmodes()
    ts.generate_mmodes_svd()
    thr = float(g["kl_threshold"])
    ts.set_kltransform("kl", threshold=thr)
    ts.generate_mmodes_kl()
    sc = np.abs(g["alm_full"]).max()
    for mi in range(lmax + 1):
        mm = ts.mmode(mi)
        ref = g["m%d_mmode" % mi]
        assert mm.shape == ref.shape and np.abs(mm - ref).max() <= 1e-13 * np.abs(ref).max(), mi
        a = bt.project_vector_telescope_to_sky(mi, mm)
        assert np.abs(a - g["alm_full"][..., mi]).max() <= 1e-9 * sc, mi
        sv = ts.mmode_svd(mi)
        assert abs(np.linalg.norm(sv) - g["svd_norm"][mi]) <= 1e-9 * max(g["svd_norm"][mi], 1e-300), mi
        a = bt.project_vector_svd_to_sky(mi, sv)
        assert np.abs(a - g["alm_svd"][..., mi]).max() <= 1e-8 * np.abs(g["alm_svd"]).max(), mi
        klm = ts.mmode_kl(mi)
        assert klm.size == int(g["nkl"][mi]), mi
        if mi >= 1:
            a = bt.project_vector_svd_to_sky(mi, kl.project_vector_kl_to_svd(mi, klm.copy(), threshold=thr))
            assert np.abs(a - g["alm_kl"][..., mi]).max() <= 1e-7 * np.abs(g["alm_kl"]).max(), mi
            ev = kl.evals_m(mi, thr)
            kw = klm.copy()
            if ev is not None:
                kw *= ev / (1.0 + ev)
            a = bt.project_vector_svd_to_sky(mi, kl.project_vector_kl_to_svd(mi, kw, threshold=thr))
            assert np.abs(a - g["alm_kl_wiener"][..., mi]).max() <= 1e-7 * np.abs(g["alm_kl_wiener"]).max(), mi
