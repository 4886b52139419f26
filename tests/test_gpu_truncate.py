"""Bit truncation of the beam-transfer blocks on the device (drift/core/beamtransfer.py:641-646) against the
oracle restatement — bit for bit, it is integer work on the mantissa — and the product path with
`truncate = True`: truncated blocks in real, chunked, lzf-compressed HDF5 files."""
import os

import numpy as np
import pytest

pytestmark = pytest.mark.gpu


def _rows(rng, nrows, ncols):
    z = (rng.standard_normal((nrows, ncols)) + 1j * rng.standard_normal((nrows, ncols))) * np.exp(-np.arange(ncols) / 9.0)
    z *= 10.0 ** rng.uniform(-12, 6, (nrows, 1))
    return z


def test_truncate_kernel_bit_exact_against_oracle():
    from driftscan_amd import device
    from oracle import truncate as ot

    device.reset_context()
    ctx = device.get_context(workspace_bytes=1 << 28)
    rng = np.random.default_rng(11)
    for nrows, ncols in ((1, 1), (7, 3), (300, 129), (64, 513), (5, 1025)):
        z = _rows(rng, nrows, ncols)
        z[:, : ncols // 5] = 0.0                      # l < m
        if nrows > 4:
            z[1] = 0.0                                 # an all-zero row
            z[2, ncols // 2] = 5e-324 + 3e-310j        # denormals
            z[3] *= 1e300                              # |z|^2 overflows: the bound becomes inf, the row collapses
            z[4, -1] = complex(np.inf, np.nan)
        for prec, pmax in ((1e-7, 1e-8), (1e-3, 0.0), (0.0, 1e-4), (0.0, 0.0), (0.5, 0.5)):
            want = ot.bit_truncate_max_complex(z, prec, pmax)
            d = ctx.to_device(z)
            ctx.bit_truncate_max_complex(d, prec, pmax)
            ctx.sync()
            got = d.cpu().numpy()
            same = (got.view(np.uint64) == want.view(np.uint64)) | (np.isnan(got.view(np.float64)) & np.isnan(want.view(np.float64)))
            assert same.all(), (nrows, ncols, prec, pmax, np.argwhere(~same)[:5])
    # the contract, at the full size of a config-3 row set
    z = _rows(rng, 4096, 513)
    d = ctx.to_device(z)
    ctx.bit_truncate_max_complex(d, 1e-7, 1e-8)
    ctx.sync()
    t = d.cpu().numpy()
    err = np.maximum(1e-7 * np.abs(z), 1e-8 * np.abs(z).max(axis=1, keepdims=True))
    assert (np.abs(t.real - z.real) <= err).all() and (np.abs(t.imag - z.imag) <= err).all()
    d2 = ctx.to_device(t)
    ctx.bit_truncate_max_complex(d2, 1e-7, 1e-8)
    ctx.sync()
    assert np.array_equal(d2.cpu().numpy(), t)        # idempotent
    tz = t.real.copy().view(np.uint64)
    low = (tz & np.uint64((1 << 24) - 1)) == 0
    assert low.mean() > 0.95                           # >= 24 trailing zero mantissa bits nearly everywhere


def test_generate_with_truncation_writes_compressed_hdf5(tmp_path, monkeypatch):
    from driftscan_amd import beamtransfer, cylinder, device, storage
    from oracle import truncate as ot

    if storage.load_driftio() is None and not storage.HAVE_H5PY:
        pytest.skip("no HDF5 library")
    monkeypatch.setenv("DRIFTMI_STORAGE", "hdf5")
    device.reset_context()
    tcfg = dict(num_freq=3, freq_start=400.0, freq_end=430.0, freq_mode="edge", num_cylinders=2, cylinder_width=3.0,
                num_feeds=4, feed_spacing=0.4, tsys=1.0)
    out = {}
    for tag, trunc in (("full", False), ("trunc", True)):
        tel = cylinder.PolarisedCylinderTelescope.from_config(dict(tcfg))
        bt = beamtransfer.BeamTransfer.from_config(dict(truncate=trunc, truncate_rel=1e-5, truncate_maxl=1e-6), str(tmp_path / tag),
                                                   telescope=tel)
        bt.generate(skip_svd=(tag == "full"))
        out[tag] = (tel, bt)
    tel, bt = out["trunc"]
    L = tel.lmax + 1
    sizes = {}
    for mi in (0, 3, tel.mmax):
        full = out["full"][1].beam_m(mi)
        want = ot.bit_truncate_max_complex(full.reshape(-1, L), 1e-5, 1e-6).reshape(full.shape)
        got = bt.beam_m(mi)
        assert np.array_equal(got, want), mi
        assert not np.array_equal(got, full) or not full.any()
        path = bt._mfile(mi)
        assert open(path, "rb").read(8) == b"\x89HDF\r\n\x1a\n"
        with storage.File(path, "r") as f:
            d = f["beam_m"]
            assert d.shape == (tel.nfreq, 2, tel.nbase, tel.num_pol_sky, L - mi)
            assert tuple(d.chunks) == (1, 2, min(10, tel.nbase), tel.num_pol_sky, L - mi) and d.compression == "lzf"
            assert int(f.attrs["m"]) == mi
        sizes[mi] = (os.path.getsize(path), os.path.getsize(out["full"][1]._mfile(mi)))
    assert sizes[0][0] < 0.75 * sizes[0][1], sizes      # truncation is what makes the files compress
    # the SVD stage ran on the truncated blocks, as the reference's does (it reads the files back)
    with storage.File(bt._svdfile(3), "r") as f:
        assert f["beam_svd"].compression == "lzf" and f["beam_svd"].chunks[0] == 1
        assert f["singularvalues"].shape == (tel.nfreq, bt.svd_len)
