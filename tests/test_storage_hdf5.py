"""CPU-only: the product files are REAL HDF5 with the reference's layout.

  * libdriftio (include/driftio.h) loads and exports what its header declares;
  * its LZF codec round-trips and rejects what does not shrink;
  * files written through driftscan_amd.storage carry the reference's dataset names, the compound {r, i}
    complex type, the chunk shapes of drift/core/beamtransfer.py:566-571 / :741-792 and lzf compression — checked
    by reading them with PLAIN h5py under /opt/conda/bin/python3.9 (the only h5py in this image), and the
    other way round: an lzf file written by h5py is read here;
  * writes go to a temporary name and are renamed (beamtransfer.py:738), several writer threads at once.
"""
import ctypes
import json
import os
import re
import subprocess
import threading

import numpy as np
import pytest

from driftscan_amd import storage

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CONDA_PY = "/opt/conda/bin/python3.9"

pytestmark = pytest.mark.skipif(storage.load_driftio() is None, reason="libdriftio / HDF5 C library not available")


def test_driftio_header_symbols_exported():
    txt = open(os.path.join(ROOT, "include", "driftio.h")).read()
    txt = re.sub(r"/\*.*?\*/", "", txt, flags=re.S)
    names = sorted(set(re.findall(r"\b(dio_[a-z0-9_]+)\s*\(", txt)))
    lib = storage.load_driftio()
    assert len(names) >= 15
    for n in names:
        assert hasattr(lib, n), "libdriftio.so does not export %s" % n
    assert sorted(storage.DIO_SIGNATURES) == names
    assert lib.dio_hdf5_version() >= 11000 and lib.dio_version() >= 100


def test_lzf_codec_roundtrip():
    lib = storage.load_driftio()
    rng = np.random.default_rng(0)
    cases = [np.zeros(5000, dtype=np.uint8), np.arange(70000, dtype=np.uint32).view(np.uint8),
             rng.integers(0, 4, 33333, dtype=np.uint8), np.frombuffer(b"abcabcabcabd" * 999, dtype=np.uint8),
             np.array([7], dtype=np.uint8), np.array([1, 1, 1], dtype=np.uint8)]
    for raw in cases:
        raw = np.ascontiguousarray(raw)
        out = np.empty(raw.size + 64, dtype=np.uint8)
        n = lib.dio_lzf_compress(raw.ctypes.data, raw.size, out.ctypes.data, out.size)
        assert n > 0
        back = np.empty(raw.size, dtype=np.uint8)
        m = lib.dio_lzf_decompress(out.ctypes.data, n, back.ctypes.data, back.size)
        assert m == raw.size and np.array_equal(back, raw)
        if raw.size > 1000:
            assert n < raw.size
        # a too-small output buffer is reported, not overrun
        assert lib.dio_lzf_decompress(out.ctypes.data, n, back.ctypes.data, max(raw.size - 1, 0)) == 0
    noise = rng.integers(0, 256, 20000, dtype=np.uint8)
    out = np.empty(noise.size, dtype=np.uint8)
    assert lib.dio_lzf_compress(noise.ctypes.data, noise.size, out.ctypes.data, noise.size - 1) == 0   # does not shrink


def _product_like(rng, F=3, B=13, P=4, L=33, mi=5, K=20):
    from oracle import truncate as ot

    bm = rng.standard_normal((F, 2, B, P, L - mi)) + 1j * rng.standard_normal((F, 2, B, P, L - mi))
    bm = ot.bit_truncate_max_complex(bm.reshape(-1, L - mi), 1e-7, 1e-8).reshape(bm.shape)   # compressible, like the real files
    return dict(beam_m=bm, beam_svd=rng.standard_normal((F, K, P, L)) + 1j * rng.standard_normal((F, K, P, L)),
                beam_ut=rng.standard_normal((F, K, 2 * B)) + 0j, singularvalues=np.abs(rng.standard_normal((F, K))))


def test_files_are_hdf5_with_reference_layout(tmp_path, monkeypatch):
    monkeypatch.setenv("DRIFTMI_STORAGE", "hdf5")
    rng = np.random.default_rng(3)
    d = _product_like(rng)
    F, _, B, P, Lm = d["beam_m"].shape
    K, L = d["beam_svd"].shape[1], d["beam_svd"].shape[3]
    pm, ps, pe = str(tmp_path / "beam.hdf5"), str(tmp_path / "svd.hdf5"), str(tmp_path / "ev_m_005.hdf5")
    with storage.File(pm, "w") as f:
        f.create_dataset("beam_m", data=d["beam_m"], **storage.compression_kwargs((1, 2, min(10, B), P, Lm)))
        f.attrs["m"] = 5
        f.attrs["frequencies"] = np.linspace(400.0, 450.0, F)
        assert not os.path.exists(pm)            # nothing under the final name until close
    assert os.path.exists(pm) and not [x for x in os.listdir(str(tmp_path)) if ".tmp" in x]
    with storage.File(ps, "w") as f:
        f.create_dataset("beam_svd", data=d["beam_svd"], **storage.compression_kwargs((1, min(10, K), P, L)))
        f.create_dataset("beam_ut", data=d["beam_ut"], **storage.compression_kwargs((1, min(10, K), 2 * B)))
        f.create_dataset("singularvalues", data=d["singularvalues"])
        f.attrs["baselines"] = rng.standard_normal((B, 2))
    with storage.File(pe, "w") as f:
        f.attrs["m"] = 5
        f.attrs["SUBSET"] = True
        f.create_dataset("evals_full", data=np.arange(7.0))
        f.create_dataset("evals", data=np.zeros(0))
        f.create_dataset("evecs", data=np.zeros((0, 7), dtype=np.complex128))
        f.attrs["num_modes"] = 0
        f.attrs["FLAGS"] = "Normal"
    for p in (pm, ps, pe):
        assert open(p, "rb").read(8) == b"\x89HDF\r\n\x1a\n"
    assert os.path.getsize(pm) < 0.8 * d["beam_m"].nbytes                    # truncated blocks do compress
    # our own reader: whole datasets, one frequency (hyperslab), slices, attributes
    with storage.File(pm, "r") as f:
        assert np.array_equal(f["beam_m"][:], d["beam_m"]) and np.array_equal(f["beam_m"][1], d["beam_m"][1])
        assert np.array_equal(f["beam_m"][1:3, 0], d["beam_m"][1:3, 0])
        assert f["beam_m"].chunks == (1, 2, 10, P, Lm) and f["beam_m"].compression == "lzf"
        assert int(f.attrs["m"]) == 5 and np.allclose(f.attrs["frequencies"], np.linspace(400.0, 450.0, F))
    with storage.File(pe, "r") as f:
        assert f["evecs"].shape == (0, 7) and f["evals"].shape == (0,) and bool(f.attrs["SUBSET"]) is True
        assert str(f.attrs["FLAGS"]) == "Normal" and int(f.attrs["num_modes"]) == 0
    if not os.path.exists(CONDA_PY):
        pytest.skip("no interpreter with h5py in this image")
    code = r'''
import h5py, json, sys, numpy as np
out = {}
with h5py.File(sys.argv[1], "r") as f:
    d = f["beam_m"]
    out["beam_m"] = dict(shape=list(d.shape), dtype=str(d.dtype), fields=list(d.id.get_type().get_member_name(i).decode() for i in range(2)),
                         chunks=list(d.chunks), compression=d.compression, sum=[float(d[:].real.sum()), float(d[:].imag.sum())],
                         m=int(f.attrs["m"]), freq=f.attrs["frequencies"].tolist())
with h5py.File(sys.argv[2], "r") as f:
    out["svd"] = {k: dict(shape=list(f[k].shape), chunks=list(f[k].chunks) if f[k].chunks else None, compression=f[k].compression,
                          dtype=str(f[k].dtype), sum=float(np.abs(f[k][:]).sum())) for k in f}
    out["svd_attrs"] = sorted(f.attrs)
with h5py.File(sys.argv[3], "r") as f:
    out["ev"] = dict(keys=sorted(f), flags=f.attrs["FLAGS"], subset=bool(f.attrs["SUBSET"]), subset_type=str(type(f.attrs["SUBSET"])),
                     nm=int(f.attrs["num_modes"]), evecs=list(f["evecs"].shape), evecs_dtype=str(f["evecs"].dtype))
with h5py.File(sys.argv[4], "w") as g:   # and a file written by h5py, for the reader of this repository
    a = np.arange(2 * 3 * 40, dtype=np.float64).reshape(2, 3, 40) * (1 + 0.5j)
    g.create_dataset("beam_m", data=a, chunks=(1, 3, 40), compression="lzf")
    g.create_dataset("singularvalues", data=np.arange(6.0).reshape(2, 3))
    g.attrs["m"] = 3
    g.attrs["FLAGS"] = "NotPositiveDefinite"
    g.attrs["add_const"] = 0.25
print(json.dumps(out))
'''
    theirs = str(tmp_path / "h5py_written.hdf5")
    res = subprocess.run([CONDA_PY, "-W", "ignore", "-c", code, pm, ps, pe, theirs], capture_output=True, text=True, timeout=300)
    assert res.returncode == 0, res.stderr
    out = json.loads(res.stdout.strip().splitlines()[-1])
    bm = out["beam_m"]
    assert bm["shape"] == list(d["beam_m"].shape) and bm["dtype"] == "complex128" and bm["fields"] == ["r", "i"]
    assert bm["chunks"] == [1, 2, 10, P, Lm] and bm["compression"] == "lzf" and bm["m"] == 5
    assert np.allclose(bm["sum"], [d["beam_m"].real.sum(), d["beam_m"].imag.sum()], rtol=1e-13)
    assert out["svd"]["beam_svd"]["chunks"] == [1, 10, P, L] and out["svd"]["beam_svd"]["compression"] == "lzf"
    assert out["svd"]["beam_ut"]["chunks"] == [1, 10, 2 * B] and out["svd"]["singularvalues"]["chunks"] is None
    assert out["svd"]["singularvalues"]["dtype"] == "float64" and out["svd_attrs"] == ["baselines"]
    for k in ("beam_svd", "beam_ut", "singularvalues"):
        assert np.isclose(out["svd"][k]["sum"], np.abs(d[k]).sum(), rtol=1e-13)
    ev = out["ev"]
    assert ev["keys"] == ["evals", "evals_full", "evecs"] and ev["flags"] == "Normal" and ev["subset"] is True
    assert "bool" in ev["subset_type"] and ev["nm"] == 0 and ev["evecs"] == [0, 7] and ev["evecs_dtype"] == "complex128"
    with storage.File(theirs, "r") as f:      # h5py's lzf chunks through our filter
        a = np.arange(2 * 3 * 40, dtype=np.float64).reshape(2, 3, 40) * (1 + 0.5j)
        assert np.array_equal(f["beam_m"][:], a) and np.array_equal(f["beam_m"][1], a[1])
        assert f["beam_m"].compression == "lzf" and f["beam_m"].chunks == (1, 3, 40)
        assert int(f.attrs["m"]) == 3 and f.attrs["FLAGS"] == "NotPositiveDefinite" and float(f.attrs["add_const"]) == 0.25


def test_concurrent_writers_and_read_modify_write(tmp_path, monkeypatch):
    monkeypatch.setenv("DRIFTMI_STORAGE", "hdf5")
    rng = np.random.default_rng(4)
    arrs = [np.round(rng.standard_normal((4, 6, 50)), 2) + 0j for _ in range(12)]
    errs = []

    def work(i):
        try:
            with storage.File(str(tmp_path / ("f%d.hdf5" % i)), "w") as f:
                f.create_dataset("beam_m", data=arrs[i], **storage.compression_kwargs((1, 6, 50)))
                f.attrs["m"] = i
        except Exception as e:  # pragma: no cover
            errs.append(e)

    ts = [threading.Thread(target=work, args=(i,)) for i in range(12)]
    [t.start() for t in ts]
    [t.join() for t in ts]
    assert not errs
    for i in range(12):
        with storage.File(str(tmp_path / ("f%d.hdf5" % i)), "r") as f:
            assert np.array_equal(f["beam_m"][:], arrs[i]) and int(f.attrs["m"]) == i
    p = str(tmp_path / "f0.hdf5")
    with storage.File(p, "r+") as f:
        f["beam_m"][2] = 7.0
        f.attrs["extra"] = "x"
    with storage.File(p, "r") as f:
        assert np.all(f["beam_m"][2] == 7.0) and np.array_equal(f["beam_m"][1], arrs[0][1]) and f.attrs["extra"] == "x"
        assert f["beam_m"].compression == "lzf"    # layout survives a read-modify-write


def test_npz_files_still_open(tmp_path, monkeypatch):
    """Products written by the .npz mirror (round 1, or a box without HDF5) are read by what they are."""
    monkeypatch.setenv("DRIFTMI_STORAGE", "npz")
    p = str(tmp_path / "old.hdf5")
    with storage.File(p, "w") as f:
        f.create_dataset("evals", data=np.arange(3.0), chunks=(3,), compression="lzf")
        f.attrs["m"] = 2
    monkeypatch.setenv("DRIFTMI_STORAGE", "hdf5")
    with storage.File(p, "r") as f:
        assert isinstance(f, storage.NpzFile) and np.array_equal(f["evals"][:], np.arange(3.0)) and int(f.attrs["m"]) == 2


@pytest.mark.parametrize("backend", ["hdf5", "npz"])
def test_failed_body_leaves_no_file_and_attrs_persist(tmp_path, monkeypatch, backend):
    """A with-body that raises writes nothing (no partial product under the final name: every stage resumes with
    "skip if the file exists"); `f.attrs[k] = v` on a file opened "r+" is persisted even when no dataset is touched,
    as h5py does."""
    monkeypatch.setenv("DRIFTMI_STORAGE", backend)
    if backend == "hdf5" and not storage.HAVE_H5PY and storage.load_driftio() is None:
        pytest.skip("no HDF5 back-end in this environment")
    p = str(tmp_path / "svd.hdf5")
    with pytest.raises(RuntimeError):
        with storage.File(p, "w") as f:
            f.create_dataset("beam_svd", data=np.arange(6.0).reshape(2, 3))
            raise RuntimeError("the stage failed between two datasets")
    assert not os.path.exists(p) and not [x for x in os.listdir(str(tmp_path)) if x.startswith("svd.hdf5")]
    with storage.File(p, "w") as f:
        f.create_dataset("beam_svd", data=np.arange(6.0).reshape(2, 3))
        f.attrs["m"] = 3
    with storage.File(p, "r+") as f:
        f.attrs["note"] = 7
    with storage.File(p, "r") as f:
        assert int(f.attrs["m"]) == 3 and int(f.attrs["note"]) == 7
        assert np.array_equal(f["beam_svd"][:], np.arange(6.0).reshape(2, 3))
    # a failing read-modify-write keeps the old content
    with pytest.raises(RuntimeError):
        with storage.File(p, "r+") as f:
            f.attrs["note"] = 9
            raise RuntimeError("no")
    with storage.File(p, "r") as f:
        assert int(f.attrs["note"]) == 7


# ---- bitshuffle + LZ4 (HDF5 filter 32008: what the reference writes when `truncate` is on) -----------------------------
def _dio_buf(lib, fn, src, cap, *extra):
    out = (ctypes.c_ubyte * max(cap, 1))()
    got = fn(bytes(src), len(src), *extra, out, cap) if extra else fn(bytes(src), len(src), out, cap)
    return bytes(out[:got]), got


def test_bitshuffle_lz4_codec_roundtrip_and_edges():
    lib = storage.load_driftio()
    rng = np.random.default_rng(11)
    # LZ4 blocks: incompressible, highly compressible, short, long matches, empty-ish
    for data in (rng.integers(0, 256, 5000, dtype=np.uint8).tobytes(), bytes(70000), b"abcd" * 3000 + b"xyz", b"q" * 11,
                 b"0123456789ab" * 2, rng.integers(0, 4, 9000, dtype=np.uint8).tobytes()):
        enc, got = _dio_buf(lib, lib.dio_lz4_compress, data, len(data) + len(data) // 255 + 64)
        assert got > 0
        dec, n = _dio_buf(lib, lib.dio_lz4_decompress, enc, len(data))
        assert n == len(data) and dec == data
    assert lib.dio_lz4_decompress(b"\xf0", 1, (ctypes.c_ubyte * 8)(), 8) == 0          # truncated length
    assert lib.dio_lz4_decompress(b"\x00\x01\x00", 3, (ctypes.c_ubyte * 8)(), 8) == 0  # offset beyond the output
    # the bit transpose and its inverse, every element size the products use
    for es, n in ((16, 512), (8, 64), (4, 8), (1, 24)):
        a = rng.integers(0, 256, n * es, dtype=np.uint8).tobytes()
        t = (ctypes.c_ubyte * (n * es))()
        b = (ctypes.c_ubyte * (n * es))()
        assert lib.dio_bitshuffle(a, t, n, es, 0) == n * es and lib.dio_bitshuffle(bytes(t), b, n, es, 1) == n * es
        assert bytes(b) == a
    assert lib.dio_bitshuffle(b"x" * 7, (ctypes.c_ubyte * 7)(), 7, 1, 0) == 0   # not a multiple of eight elements
    # whole chunks: full blocks + a shorter block + a tail of fewer than eight elements
    for es, n in ((16, 512 * 3 + 77), (8, 5), (16, 512), (4, 2048 * 2 + 8)):
        a = (rng.integers(0, 3, n * es, dtype=np.uint8) * 5).tobytes()
        cap = n * es + 4096
        out = (ctypes.c_ubyte * cap)()
        got = lib.dio_bshuf_lz4_encode(a, len(a), es, 0, out, cap)
        assert got >= 12
        hdr = bytes(out[:12])
        assert int.from_bytes(hdr[:8], "big") == len(a) and int.from_bytes(hdr[8:], "big") == (8192 // es // 8 * 8) * es
        back = (ctypes.c_ubyte * len(a))()
        assert lib.dio_bshuf_lz4_decode(bytes(out[:got]), got, es, back, len(a)) == len(a) and bytes(back) == a
        assert lib.dio_bshuf_lz4_decode(bytes(out[: got - 1]), got - 1, es, back, len(a)) == 0 or n * es % (8 * es) != 0


def test_bitshuffle_layout_against_its_statement():
    """The vectorised forward transform (16 x 16 byte transposes + movemask bit rows, csrc/dm_h5io.c) against the layout
    written out with numpy: bit k of byte b of the elements of a block, eight elements to a byte (bit j from element
    8 i + j), rows ordered (b, k); blocks as bshuf_bitshuffle cuts them; sizes that exercise the 32-, 16- and 8-wide
    steps and the element sizes without a vector byte stage."""
    lib = storage.load_driftio()
    rng = np.random.default_rng(1)

    def statement(raw, n, es, blk):
        out = np.empty_like(raw)
        done = 0
        while done < n:
            cur = blk if n - done >= blk else ((n - done) // 8) * 8
            if cur == 0:
                break
            x = raw[done * es : (done + cur) * es].reshape(cur, es)
            bits = np.unpackbits(x.T.reshape(es, cur, 1), axis=2, bitorder="little")
            out[done * es : (done + cur) * es] = np.packbits(bits.transpose(0, 2, 1).reshape(es * 8, cur), axis=1,
                                                              bitorder="little").ravel()
            done += cur
        out[done * es :] = raw[done * es :]
        return out

    for es in (1, 3, 4, 8, 16):
        for n in (8, 16, 24, 40, 512, 520, 1027, 4099):
            for blk in (0, 8, 64, 1024):
                raw = rng.integers(0, 256, n * es, dtype=np.uint8)
                out, back = np.empty_like(raw), np.empty_like(raw)
                b = blk if blk else max(8, (8192 // es) // 8 * 8)
                assert lib.dio_bitshuffle_blocked(raw.ctypes.data, out.ctypes.data, n, es, blk, 0) == n * es
                assert np.array_equal(out, statement(raw, n, es, b)), (es, n, blk)
                assert lib.dio_bitshuffle_blocked(out.ctypes.data, back.ctypes.data, n, es, blk, 1) == n * es
                assert np.array_equal(back, raw)


@pytest.mark.skipif(not os.path.exists(CONDA_PY), reason="no interpreter with imagecodecs")
def test_bitshuffle_and_lz4_against_the_real_libraries(tmp_path):
    """The bit transpose against the bitshuffle library and the LZ4 blocks against liblz4, both through imagecodecs under
    /opt/conda (it wraps the two C libraries).  The HDF5 chunk framing around them follows bshuf_h5filter.c and has no
    counterpart in this image (no bitshuffle HDF5 plugin): unpinned."""
    probe = subprocess.run([CONDA_PY, "-c", "import imagecodecs; imagecodecs.bitshuffle_encode; imagecodecs.lz4_encode"],
                           capture_output=True)
    if probe.returncode != 0:
        pytest.skip("imagecodecs with bitshuffle / lz4 not importable")
    lib = storage.load_driftio()
    rng = np.random.default_rng(5)
    a = (rng.standard_normal(512) + 1j * rng.standard_normal(512)).astype(np.complex128)
    a.real = np.round(a.real * 64) / 64
    a.imag = np.round(a.imag * 64) / 64          # truncated-looking values: the low mantissa bits are zero
    raw = a.tobytes()
    mine = (ctypes.c_ubyte * len(raw))()
    assert lib.dio_bitshuffle(raw, mine, 512, 16, 0) == len(raw)
    enc, got = _dio_buf(lib, lib.dio_lz4_compress, bytes(mine), len(raw) + 256)
    np.save(tmp_path / "a.npy", a)
    (tmp_path / "mine_lz4.bin").write_bytes(enc)
    script = (
        "import sys, numpy as np, imagecodecs\n"
        "d = sys.argv[1]\n"
        "a = np.load(d + '/a.npy')\n"
        "ref = imagecodecs.bitshuffle_encode(a, blocksize=512)   # element size from the dtype: 16\n"
        "open(d + '/ref_shuf.bin', 'wb').write(bytes(ref))\n"
        "open(d + '/ref_lz4.bin', 'wb').write(bytes(imagecodecs.lz4_encode(bytes(ref), header=False)))\n"
        "back = imagecodecs.lz4_decode(open(d + '/mine_lz4.bin', 'rb').read(), header=False, out=len(bytes(ref)))\n"
        "open(d + '/mine_decoded_by_liblz4.bin', 'wb').write(bytes(back))\n")
    res = subprocess.run([CONDA_PY, "-c", script, str(tmp_path)], capture_output=True, text=True)
    assert res.returncode == 0, res.stderr
    ref_shuf = (tmp_path / "ref_shuf.bin").read_bytes()
    assert ref_shuf == bytes(mine)                                        # the transform of the bitshuffle library
    assert (tmp_path / "mine_decoded_by_liblz4.bin").read_bytes() == bytes(mine)   # liblz4 reads this encoder's block
    ref_lz4 = (tmp_path / "ref_lz4.bin").read_bytes()
    dec, n = _dio_buf(lib, lib.dio_lz4_decompress, ref_lz4, len(raw))
    assert n == len(raw) and dec == ref_shuf                              # this decoder reads liblz4's block
    assert got < len(raw) // 2                                            # and the truncated block does shrink
    # the blocking of a whole buffer (three default blocks of 512 elements, one of 72, five elements copied) against
    # bshuf_bitshuffle with its default block size
    n = 512 * 3 + 77
    b = (rng.standard_normal(n) + 1j * rng.standard_normal(n)).astype(np.complex128)
    np.save(tmp_path / "b.npy", b)
    res = subprocess.run([CONDA_PY, "-c", "import sys, numpy as np, imagecodecs\nd = sys.argv[1]\n"
                          "open(d + '/ref_blocked.bin', 'wb').write(bytes(imagecodecs.bitshuffle_encode(np.load(d + '/b.npy'))))\n",
                          str(tmp_path)], capture_output=True, text=True)
    assert res.returncode == 0, res.stderr
    blocked = (ctypes.c_ubyte * (n * 16))()
    assert lib.dio_bitshuffle_blocked(b.tobytes(), blocked, n, 16, 0, 0) == n * 16
    assert bytes(blocked) == (tmp_path / "ref_blocked.bin").read_bytes()
    undone = (ctypes.c_ubyte * (n * 16))()
    assert lib.dio_bitshuffle_blocked(bytes(blocked), undone, n, 16, 0, 1) == n * 16 and bytes(undone) == b.tobytes()


def test_bitshuffle_files_roundtrip(tmp_path, monkeypatch):
    """A product dataset written with filter 32008 reads back through the library's own filter (dataset info names the codec);
    plain h5py sees the filter id and the five cd_values bitshuffle's plugin leaves in a file."""
    monkeypatch.setenv("DRIFTMI_STORAGE", "hdf5")
    monkeypatch.setenv("DRIFTMI_H5_CODEC", "bitshuffle")
    rng = np.random.default_rng(2)
    beam = np.round((rng.standard_normal((3, 2, 7, 1, 41)) + 1j * rng.standard_normal((3, 2, 7, 1, 41))) * 256) / 256
    path = str(tmp_path / "beam.hdf5")
    with storage.File(path, "w") as f:
        f.create_dataset("beam_m", data=beam, **storage.compression_kwargs((1, 2, 7, 1, 41)))
        f.create_dataset("dense", data=rng.standard_normal((5, 300)), **storage.compression_kwargs((1, 300)))
    with storage.File(path, "r") as f:
        assert f["beam_m"].compression == "bitshuffle" and f["beam_m"].chunks == (1, 2, 7, 1, 41)
        assert np.array_equal(f["beam_m"][:], beam)
        assert np.array_equal(f["beam_m"][1, :, 2:5], beam[1, :, 2:5])
        assert f["dense"][:].shape == (5, 300)
    if os.path.exists(CONDA_PY):
        script = ("import h5py, json, sys\n"
                  "f = h5py.File(sys.argv[1], 'r')\n"
                  "pl = f['beam_m'].id.get_create_plist()\n"
                  "print(json.dumps([pl.get_nfilters(), list(pl.get_filter(0)[:1]), list(pl.get_filter(0)[2])]))\n")
        res = subprocess.run([CONDA_PY, "-c", script, path], capture_output=True, text=True)
        assert res.returncode == 0, res.stderr
        nf, fid, cd = json.loads(res.stdout)
        assert nf == 1 and fid == [32008] and cd == [0, 4, 16, 0, 2]


def test_writer_processes_produce_the_same_files(tmp_path, monkeypatch):
    """DRIFTMI_IO_PROCS: the writer threads hand an assembled file to worker PROCESSES (each with its own libhdf5; dataset
    bytes through POSIX shared memory) — same bytes, same layout, attributes and chunking as the in-process path, errors
    surface in flush(), nothing is left in /dev/shm."""
    monkeypatch.setenv("DRIFTMI_STORAGE", "hdf5")
    monkeypatch.setenv("DRIFTMI_IO_THREADS", "4")
    rng = np.random.default_rng(3)
    data = {i: rng.standard_normal((6, 2, 10, 4, 300)) + 1j * rng.standard_normal((6, 2, 10, 4, 300)) for i in range(5)}

    def write(tag, i):
        with storage.File(str(tmp_path / ("%s_%d.hdf5" % (tag, i))), "w") as f:
            f.create_dataset("beam_m", data=data[i], **storage.compression_kwargs((1, 2, 10, 4, 300)))
            f.create_dataset("singularvalues", data=np.arange(7.0) + i)
            f.create_dataset("empty", data=np.zeros((0, 3)))
            f.attrs["m"] = i
            f.attrs["frequencies"] = np.linspace(400.0, 450.0, 6)
            f.attrs["FLAGS"] = "Normal"
            f.attrs["SUBSET"] = True

    shm_before = set(os.listdir("/dev/shm")) if os.path.isdir("/dev/shm") else set()
    try:
        for tag, procs in (("thr", "0"), ("proc", "3")):
            monkeypatch.setenv("DRIFTMI_IO_PROCS", procs)
            for i in data:
                storage.submit(write, tag, i)
            storage.flush()
        for i in data:
            pa, pb = str(tmp_path / ("thr_%d.hdf5" % i)), str(tmp_path / ("proc_%d.hdf5" % i))
            assert os.path.getsize(pa) == os.path.getsize(pb)
            with storage.File(pa, "r") as a, storage.File(pb, "r") as b:
                assert np.array_equal(a["beam_m"][:], data[i]) and np.array_equal(b["beam_m"][:], data[i])
                assert np.array_equal(a["singularvalues"][:], b["singularvalues"][:]) and b["empty"].shape == (0, 3)
                assert a["beam_m"].chunks == b["beam_m"].chunks == (1, 2, 10, 4, 300)
                assert a["beam_m"].compression == b["beam_m"].compression == "lzf"
                assert b.attrs["m"] == i and b.attrs["FLAGS"] == "Normal" and bool(b.attrs["SUBSET"])
                assert np.array_equal(a.attrs["frequencies"], b.attrs["frequencies"])
        # a failing write in a worker process surfaces in flush(), and the pool keeps working afterwards
        monkeypatch.setenv("DRIFTMI_IO_PROCS", "3")

        def bad():
            with storage.File(str(tmp_path / "no_such_dir" / "x.hdf5"), "w") as f:
                f.create_dataset("big", data=data[0])

        storage.submit(bad)
        with pytest.raises(IOError):
            storage.flush()
        storage.submit(write, "again", 0)
        storage.flush()
        assert os.path.exists(str(tmp_path / "again_0.hdf5")) and not [p for p in os.listdir(str(tmp_path)) if ".tmp" in p]
    finally:
        storage.shutdown_writers()
    if os.path.isdir("/dev/shm"):
        assert set(os.listdir("/dev/shm")) - shm_before == set()


def test_chunk_threads_do_not_change_the_file(tmp_path, monkeypatch):
    """`dio_write_dataset` compresses blocks of chunks with DRIFTMI_IO_CHUNK_THREADS threads and hands them to HDF5 in order:
    the same datasets, the same stored sizes and — apart from the modification times in the object headers — the same bytes
    whatever the thread count; ragged edge chunks, raw-stored (incompressible) and packed chunks in one dataset, both codecs."""
    monkeypatch.setenv("DRIFTMI_STORAGE", "hdf5")
    if storage.backend() != "driftio":
        pytest.skip("libdriftio is not the writer in use")
    rng = np.random.default_rng(5)
    a = rng.standard_normal((37, 10, 4, 513)) + 1j * rng.standard_normal((37, 10, 4, 513))
    a[:, 5:] = 0                                         # zero rows, as beam_svd has beyond nmodes
    b = rng.standard_normal((300, 70, 41))
    sizes = {}
    for nt in ("1", "4", "7"):
        monkeypatch.setenv("DRIFTMI_IO_CHUNK_THREADS", nt)
        p = str(tmp_path / ("t%s.hdf5" % nt))
        with storage.File(p, "w") as f:
            f.create_dataset("a", data=a, chunks=(1, 10, 4, 513), compression="lzf")
            f.create_dataset("b", data=b, chunks=(16, 70, 41), compression="lzf")
            f.create_dataset("c", data=b, chunks=(7, 33, 41), compression="bitshuffle")
        sizes[nt] = os.path.getsize(p)
        with storage.File(p, "r") as f:
            assert np.array_equal(f["a"][...], a) and np.array_equal(f["b"][...], b) and np.array_equal(f["c"][...], b)
    assert sizes["1"] == sizes["4"] == sizes["7"]
    raw = {nt: np.frombuffer(open(str(tmp_path / ("t%s.hdf5" % nt)), "rb").read(), np.uint8) for nt in sizes}
    assert (raw["1"] != raw["4"]).sum() <= 16 and (raw["1"] != raw["7"]).sum() <= 16   # object-header time stamps only


def test_writer_pool_survives_a_dead_process_and_an_unstartable_one(tmp_path, monkeypatch):
    """The pool never shrinks: a writer process that dies is replaced in its slot; when no process can be started any more
    the slot stays as a marker and its files are written in the writer thread — flush() raises the transport error once and
    never blocks; a REPORTED write failure (`WriteFailed`) leaves the process where it is."""
    monkeypatch.setenv("DRIFTMI_STORAGE", "hdf5")
    if storage.backend() != "driftio":
        pytest.skip("libdriftio is not the writer in use")
    monkeypatch.setenv("DRIFTMI_IO_THREADS", "2")
    monkeypatch.setenv("DRIFTMI_IO_PROCS", "2")
    rng = np.random.default_rng(7)
    big = rng.standard_normal((200, 1000))

    def write(tag):
        with storage.File(str(tmp_path / ("%s.hdf5" % tag)), "w") as f:
            f.create_dataset("x", data=big)

    try:
        storage.submit(write, "a")
        storage.flush()
        assert len(storage._worker_list) == 2
        pids = sorted(w.p.pid for w in storage._worker_list)
        # a reported failure keeps both processes
        def bad():
            with storage.File(str(tmp_path / "missing" / "x.hdf5"), "w") as f:
                f.create_dataset("x", data=big)

        storage.submit(bad)
        with pytest.raises(IOError):
            storage.flush()
        assert sorted(w.p.pid for w in storage._worker_list) == pids
        # kill both processes: the next writes hit a broken pipe, fresh processes take the slots
        for w in list(storage._worker_list):
            w.p.kill(); w.p.wait()
        def drain():   # flush() re-raises the FIRST failure and leaves the rest queued: wait for all of them
            n = 0
            while True:
                try:
                    storage.flush()
                    return n
                except IOError:
                    n += 1

        storage.submit(write, "b"); storage.submit(write, "c")
        assert drain() >= 1
        assert len(storage._worker_list) == 2 and sorted(w.p.pid for w in storage._worker_list) != pids
        storage.submit(write, "d")
        storage.flush()
        assert os.path.exists(str(tmp_path / "d.hdf5"))
        # no process can be started any more: kill them again with a _Worker that refuses to start
        for w in list(storage._worker_list):
            w.p.kill(); w.p.wait()

        class Refuses(object):
            def __init__(self):
                raise OSError("no more processes")

        monkeypatch.setattr(storage, "_Worker", Refuses)
        storage.submit(write, "e"); storage.submit(write, "f")
        assert drain() >= 1
        assert storage._worker_list == []
        for tag in ("g", "h", "i"):                       # slots are markers now: written in the writer threads, no deadlock
            storage.submit(write, tag)
        storage.flush()
        for tag in ("g", "h", "i"):
            with storage.File(str(tmp_path / ("%s.hdf5" % tag)), "r") as f:
                assert np.array_equal(f["x"][...], big)
    finally:
        monkeypatch.undo()
        storage.shutdown_writers()
