"""Product files.

Three back-ends behind one small h5py-like interface (``File``, ``create_dataset``, ``f[name][idx]``,
``f.attrs``), picked in this order:

  h5py       when it is importable (not in this image's primary Python);
  libdriftio the in-tree C library (``include/driftio.h``, ``csrc/dm_h5io.c``) on the HDF5 C API: REAL
             HDF5 files with the reference's dataset names, chunk shapes, the compound ``{r, i}`` complex
             type h5py uses and LZF compression (filter 32000, its own codec), readable by plain h5py —
             and lzf files written by the reference are readable here.  Chunks are compressed by the calling
             thread outside the library's HDF5 lock, so the writer pool below scales with its threads;
  npz        a single-file ``.npz`` mirror with the same names, when no HDF5 library can be loaded.

Every back-end writes to ``<path>.tmp<pid>`` and renames on close (the reference guards its svd files
the same way, ``caput.misc.lock_file``, drift/core/beamtransfer.py:738): a killed run never leaves a
truncated file under the final name, so "skip if the file exists" resumes are safe.
``DRIFTMI_STORAGE = hdf5 | npz`` forces a back-end.  Readers look at the file's magic bytes, not at the
configured back-end.
"""
import ctypes
import os
import threading
import time

import numpy as np

try:  # pragma: no cover - depends on the environment
    import h5py  # noqa: F401

    HAVE_H5PY = True
except Exception:  # pragma: no cover
    h5py = None
    HAVE_H5PY = False

_HERE = os.path.dirname(os.path.abspath(__file__))
_DIO_PATH = os.path.join(_HERE, "lib", "libdriftio.so")
_dio = None
_dio_tried = False
_dio_lock = threading.Lock()   # the first caller may be any of the writer threads

F64, C128, I64, I32, BOOL, STR, F32, OTHER = 0, 1, 2, 3, 4, 5, 6, 99
COMP_NONE, COMP_LZF, COMP_BSHUF_LZ4 = 0, 1, 2
_NP2DIO = {np.dtype(np.float64): F64, np.dtype(np.complex128): C128, np.dtype(np.int64): I64,
           np.dtype(np.int32): I32, np.dtype(np.bool_): BOOL, np.dtype(np.float32): F32}
_DIO2NP = {v: k for k, v in _NP2DIO.items()}

DIO_SIGNATURES = {
    "dio_last_error": (ctypes.c_char_p, []),
    "dio_version": (ctypes.c_int, []),
    "dio_hdf5_version": (ctypes.c_int, []),
    "dio_open": (ctypes.c_int, [ctypes.c_char_p, ctypes.c_char_p, ctypes.POINTER(ctypes.c_int64)]),
    "dio_create": (ctypes.c_int, [ctypes.c_char_p, ctypes.c_uint64, ctypes.POINTER(ctypes.c_int64)]),
    "dio_close": (ctypes.c_int, [ctypes.c_int64]),
    "dio_write_dataset": (ctypes.c_int, [ctypes.c_int64, ctypes.c_char_p, ctypes.c_int, ctypes.c_int,
                                         ctypes.POINTER(ctypes.c_uint64), ctypes.POINTER(ctypes.c_uint64), ctypes.c_int,
                                         ctypes.c_void_p]),
    "dio_dataset_info": (ctypes.c_int, [ctypes.c_int64, ctypes.c_char_p, ctypes.POINTER(ctypes.c_int),
                                        ctypes.POINTER(ctypes.c_int), ctypes.POINTER(ctypes.c_uint64),
                                        ctypes.POINTER(ctypes.c_uint64), ctypes.POINTER(ctypes.c_int)]),
    "dio_read_dataset": (ctypes.c_int, [ctypes.c_int64, ctypes.c_char_p, ctypes.c_int, ctypes.POINTER(ctypes.c_uint64),
                                        ctypes.POINTER(ctypes.c_uint64), ctypes.c_void_p]),
    "dio_exists": (ctypes.c_int, [ctypes.c_int64, ctypes.c_char_p]),
    "dio_list": (ctypes.c_int64, [ctypes.c_int64, ctypes.c_char_p, ctypes.c_int64]),
    "dio_write_attr": (ctypes.c_int, [ctypes.c_int64, ctypes.c_char_p, ctypes.c_int, ctypes.c_int,
                                      ctypes.POINTER(ctypes.c_uint64), ctypes.c_void_p]),
    "dio_attr_info": (ctypes.c_int, [ctypes.c_int64, ctypes.c_char_p, ctypes.POINTER(ctypes.c_int),
                                     ctypes.POINTER(ctypes.c_int), ctypes.POINTER(ctypes.c_uint64),
                                     ctypes.POINTER(ctypes.c_int64)]),
    "dio_read_attr": (ctypes.c_int, [ctypes.c_int64, ctypes.c_char_p, ctypes.c_int, ctypes.c_void_p, ctypes.c_int64]),
    "dio_list_attrs": (ctypes.c_int64, [ctypes.c_int64, ctypes.c_char_p, ctypes.c_int64]),
    "dio_lzf_compress": (ctypes.c_size_t, [ctypes.c_void_p, ctypes.c_size_t, ctypes.c_void_p, ctypes.c_size_t]),
    "dio_lzf_decompress": (ctypes.c_size_t, [ctypes.c_void_p, ctypes.c_size_t, ctypes.c_void_p, ctypes.c_size_t]),
    "dio_bitshuffle": (ctypes.c_size_t, [ctypes.c_void_p, ctypes.c_void_p, ctypes.c_size_t, ctypes.c_size_t, ctypes.c_int]),
    "dio_bitshuffle_blocked": (ctypes.c_size_t, [ctypes.c_void_p, ctypes.c_void_p, ctypes.c_size_t, ctypes.c_size_t,
                                                 ctypes.c_size_t, ctypes.c_int]),
    "dio_lz4_compress": (ctypes.c_size_t, [ctypes.c_void_p, ctypes.c_size_t, ctypes.c_void_p, ctypes.c_size_t]),
    "dio_lz4_decompress": (ctypes.c_size_t, [ctypes.c_void_p, ctypes.c_size_t, ctypes.c_void_p, ctypes.c_size_t]),
    "dio_bshuf_lz4_encode": (ctypes.c_size_t, [ctypes.c_void_p, ctypes.c_size_t, ctypes.c_size_t, ctypes.c_size_t,
                                               ctypes.c_void_p, ctypes.c_size_t]),
    "dio_bshuf_lz4_decode": (ctypes.c_size_t, [ctypes.c_void_p, ctypes.c_size_t, ctypes.c_size_t, ctypes.c_void_p,
                                               ctypes.c_size_t]),
}


def load_driftio():
    """libdriftio.so with its prototypes declared, or None when it (or the HDF5 library it links) is absent."""
    global _dio, _dio_tried
    with _dio_lock:
        if _dio_tried:
            return _dio
        try:
            lib = ctypes.CDLL(_DIO_PATH)
            for name, (res, args) in DIO_SIGNATURES.items():
                fn = getattr(lib, name)
                fn.restype, fn.argtypes = res, args
            lib.dio_hdf5_version()
            _dio = lib
        except (OSError, AttributeError):
            _dio = None
        _dio_tried = True
        return _dio


def backend():
    forced = os.environ.get("DRIFTMI_STORAGE", "").lower()
    if forced in ("npz", "discard"):
        return "npz"
    if forced == "hdf5":
        if HAVE_H5PY:
            return "h5py"
        if load_driftio() is None:
            raise IOError("DRIFTMI_STORAGE=hdf5 but neither h5py nor libdriftio (HDF5 C library) can be loaded")
        return "driftio"
    if HAVE_H5PY:
        return "h5py"
    if load_driftio() is not None:
        return "driftio"
    global _warned_npz
    if not _warned_npz:
        _warned_npz = True
        import warnings

        warnings.warn("driftscan_amd.storage: no HDF5 back-end (neither h5py nor libdriftio.so / libhdf5): product files are "
                      "written as .npz archives under their *.hdf5 names and are NOT readable by the reference; build "
                      "libdriftio (csrc/Makefile, needs the HDF5 C library) or set DRIFTMI_STORAGE=npz to silence this",
                      RuntimeWarning, stacklevel=2)
    return "npz"


_warned_npz = False


def _is_hdf5(path):
    try:
        with open(path, "rb") as fh:
            return fh.read(8) == b"\x89HDF\r\n\x1a\n"
    except IOError:
        return False


# ---- in-memory dataset shared by the write-at-close back-ends -----------------------------------
class _MemDataset(object):
    def __init__(self, owner, name):
        self._o, self._n = owner, name

    def _arr(self):
        return self._o._load(self._n)

    @property
    def shape(self):
        return self._o._shape(self._n)

    @property
    def dtype(self):
        return self._o._dtype(self._n)

    @property
    def chunks(self):
        return self._o._layout(self._n)[0]

    @property
    def compression(self):
        return self._o._layout(self._n)[1]

    def __getitem__(self, idx):
        part = self._o._read_part(self._n, idx)
        if part is not None:
            return part
        out = self._arr()[idx]
        return np.array(out) if isinstance(out, np.ndarray) else out

    def __setitem__(self, idx, val):
        self._arr()[idx] = val
        self._o._dirty = True

    def __len__(self):
        return self.shape[0]


class _Attrs(dict):
    """Attribute dictionary of a write-at-close file: any change marks the file dirty (h5py persists ``f.attrs[k] = v``
    on a file opened "a" / "r+" whether or not a dataset was touched)."""

    def __init__(self, owner):
        dict.__init__(self)
        self._owner = owner

    def _touch(self):
        if self._owner.mode != "r":
            self._owner._dirty = True

    def __setitem__(self, k, v):
        dict.__setitem__(self, k, v)
        self._touch()

    def __delitem__(self, k):
        dict.__delitem__(self, k)
        self._touch()

    def update(self, *a, **kw):
        dict.update(self, *a, **kw)
        self._touch()

    def pop(self, *a):
        r = dict.pop(self, *a)
        self._touch()
        return r

    def clear(self):
        dict.clear(self)
        self._touch()

    def setdefault(self, k, d=None):
        if k not in self:
            self[k] = d
        return dict.__getitem__(self, k)


class _BaseFile(object):
    """Write-at-close file: datasets live as numpy arrays until ``close`` writes them to a temporary file that is
    then renamed.  Subclasses provide ``_open_existing``, ``_read``, ``_write_all``."""

    def __init__(self, path, mode="r"):
        self.path, self.mode = path, mode
        self._data, self._opts, self._dirty = {}, {}, False
        self.attrs = _Attrs(self)
        self._h = None
        if mode in ("r", "r+", "a") and os.path.exists(path):
            self._open_existing()
            self._dirty = False   # (reading the stored attributes in is not a change)
        elif mode in ("r", "r+"):
            raise IOError("no such file: %s" % path)
        if mode == "w":
            self._dirty = True

    # -- subclass hooks
    def _open_existing(self):
        raise NotImplementedError

    def _read(self, name):
        raise NotImplementedError

    def _write_all(self, tmp):
        raise NotImplementedError

    def _close_handle(self):
        pass

    def _info(self, name):
        return None

    def _read_part(self, name, idx):
        return None

    # -- h5py-like surface
    def create_dataset(self, name, shape=None, dtype=None, data=None, chunks=None, compression=None, **kwargs):
        if data is not None:
            # no copy for arrays handed over for writing (the products are hundreds of MB per file)
            arr = np.asarray(data, dtype=dtype) if dtype is not None else np.asarray(data)
        else:
            arr = np.zeros(shape, dtype=dtype)
        self._data[name] = arr
        self._opts[name] = (tuple(int(c) for c in chunks) if chunks is not None else None, compression)
        self._dirty = True
        return _MemDataset(self, name)

    def _load(self, name):
        if self._data[name] is None:
            self._data[name] = self._read(name)
        return self._data[name]

    def _shape(self, name):
        if self._data[name] is None:
            info = self._info(name)
            if info is not None:
                return info[1]
        return self._load(name).shape

    def _dtype(self, name):
        if self._data[name] is None:
            info = self._info(name)
            if info is not None:
                return info[0]
        return self._load(name).dtype

    def _layout(self, name):
        if name in self._opts:
            return self._opts[name]
        info = self._info(name)
        return (info[2], info[3]) if info is not None else (None, None)

    def __getitem__(self, name):
        if name not in self._data:
            raise KeyError(name)
        return _MemDataset(self, name)

    def __contains__(self, name):
        return name in self._data

    def keys(self):
        return self._data.keys()

    def close(self):
        if self.mode != "r" and self._dirty:
            for k in self._data:   # read-modify-write: everything must be in memory before the old file goes away
                self._load(k)
                if k not in self._opts:
                    chunks, comp = self._layout(k)   # keep the chunk shape / compression the dataset had
                    self._opts[k] = (chunks, comp if comp in (None, "lzf", "bitshuffle") else None)
            self._close_handle()
            tmp = self.path + ".tmp%d_%d" % (os.getpid(), threading.get_ident())
            try:
                self._write_all(tmp)
                os.replace(tmp, self.path)  # write-temp-then-rename, like caput.misc.lock_file
            finally:
                if os.path.exists(tmp):
                    os.remove(tmp)
            self._dirty = False
        self._close_handle()

    def __enter__(self):
        return self

    def __exit__(self, et, ev, tb):
        if et is not None:
            # the with-body failed: nothing is written, no partial file appears under the final name (every stage of the
            # pipeline resumes with "skip if the file exists", so a half-written product would be kept for ever)
            self._dirty = False
            self._close_handle()
            return False
        self.close()
        return False


class NpzFile(_BaseFile):
    """Minimal h5py.File look-alike persisted as one ``.npz`` archive at ``path``."""

    def _open_existing(self):
        z = np.load(self.path, allow_pickle=False)
        for k in z.files:
            if k.startswith("__attr__"):
                v = z[k]
                self.attrs[k[8:]] = v.item() if v.shape == () else v
            else:
                self._data[k] = None
        self._h = z   # datasets are loaded when first touched

    def _read(self, name):
        return self._h[name]

    def _write_all(self, tmp):
        payload = {k: self._data[k] for k in self._data}
        for k, v in self.attrs.items():
            payload["__attr__" + k] = np.asarray(v)
        with open(tmp, "wb") as fh:
            np.savez(fh, **payload)

    def _close_handle(self):
        if self._h is not None:
            self._h.close()
            self._h = None


def _u64(seq):
    return (ctypes.c_uint64 * max(len(seq), 1))(*[int(x) for x in seq])


class DriftioFile(_BaseFile):
    """HDF5 through libdriftio (include/driftio.h)."""

    def __init__(self, path, mode="r"):
        self._lib = load_driftio()
        if self._lib is None:
            raise IOError("libdriftio is not available")
        self._infos = {}
        _BaseFile.__init__(self, path, mode)

    def _check(self, rc, what):
        if rc < 0:
            raise IOError("%s: %s" % (what, (self._lib.dio_last_error() or b"").decode()))
        return rc

    @staticmethod
    def _names(fn, h):
        n = fn(h, None, 0)
        if n <= 0:
            return []
        buf = ctypes.create_string_buffer(int(n) + 1)
        fn(h, buf, n)
        return [s for s in buf.raw[: int(n)].decode().split("\n") if s]

    def _open_existing(self):
        h = ctypes.c_int64(0)
        self._check(self._lib.dio_open(self.path.encode(), b"r", ctypes.byref(h)), "open %s" % self.path)
        self._h = h
        for name in self._names(self._lib.dio_list, h):
            self._data[name] = None
        for name in self._names(self._lib.dio_list_attrs, h):
            self.attrs[name] = self._read_attr(name)

    def _read_attr(self, name):
        dt, nd, sl = ctypes.c_int(0), ctypes.c_int(0), ctypes.c_int64(0)
        shape = (ctypes.c_uint64 * 8)()
        self._check(self._lib.dio_attr_info(self._h, name.encode(), ctypes.byref(dt), ctypes.byref(nd), shape,
                                            ctypes.byref(sl)), "attribute %s" % name)
        if dt.value == STR:
            buf = ctypes.create_string_buffer(int(sl.value) + 1)
            self._check(self._lib.dio_read_attr(self._h, name.encode(), STR, buf, int(sl.value) + 1), "attribute %s" % name)
            return buf.value.decode()
        if dt.value not in _DIO2NP:
            return None
        out = np.zeros(tuple(int(shape[i]) for i in range(nd.value)), dtype=_DIO2NP[dt.value])
        self._check(self._lib.dio_read_attr(self._h, name.encode(), dt.value, out.ctypes.data_as(ctypes.c_void_p), out.nbytes),
                    "attribute %s" % name)
        return out[()] if out.shape == () else out

    def _info(self, name):
        if name not in self._infos:
            if self._h is None:
                return None
            dt, nd, comp = ctypes.c_int(0), ctypes.c_int(0), ctypes.c_int(0)
            shape, chunks = (ctypes.c_uint64 * 8)(), (ctypes.c_uint64 * 8)()
            self._check(self._lib.dio_dataset_info(self._h, name.encode(), ctypes.byref(dt), ctypes.byref(nd), shape, chunks,
                                                   ctypes.byref(comp)), "dataset %s" % name)
            shp = tuple(int(shape[i]) for i in range(nd.value))
            chk = tuple(int(chunks[i]) for i in range(nd.value))
            self._infos[name] = (_DIO2NP.get(dt.value, np.dtype(np.float64)), shp, chk if any(chk) else None,
                                 {COMP_NONE: None, COMP_LZF: "lzf", COMP_BSHUF_LZ4: "bitshuffle"}.get(comp.value, "unknown"), dt.value)
        return self._infos[name]

    def _read(self, name, start=None, count=None):
        info = self._info(name)
        if info[4] not in _DIO2NP:
            raise IOError("dataset %s has an element type this reader does not handle" % name)
        shape = info[1] if count is None else tuple(count)
        out = np.empty(shape, dtype=info[0])
        if out.size:
            st = _u64(start) if start is not None else None
            ct = _u64(count) if count is not None else None
            self._check(self._lib.dio_read_dataset(self._h, name.encode(), info[4], st, ct,
                                                   out.ctypes.data_as(ctypes.c_void_p)), "read %s" % name)
        return out

    def _read_part(self, name, idx):
        """Index the leading axis with an int or a slice straight from the file (one frequency of a block:
        `beam_m(mi, fi)` of the reference reads `fh[dset][fi]`) instead of loading the whole dataset."""
        if self._data.get(name) is not None or self._h is None:
            return None
        info = self._info(name)
        shp = info[1]
        if not shp:
            return None
        first, rest = (idx[0], idx[1:]) if isinstance(idx, tuple) and idx else (idx, ())
        if isinstance(first, (int, np.integer)):
            i = int(first) + (shp[0] if first < 0 else 0)
            if not 0 <= i < shp[0]:
                raise IndexError(first)
            part = self._read(name, (i,) + (0,) * (len(shp) - 1), (1,) + shp[1:])[0]
        elif isinstance(first, slice) and first.step in (None, 1) and first != slice(None):
            a, b, _ = first.indices(shp[0])
            part = self._read(name, (a,) + (0,) * (len(shp) - 1), (max(b - a, 0),) + shp[1:])
        else:
            return None
        return part[rest] if rest else part

    def _write_all(self, tmp):
        # large new files handed over by the writer threads go to the writer PROCESSES when there are any (below)
        if self.mode == "w" and _io_procs() > 0 and threading.current_thread().name.startswith("driftmi-io") and \
                sum(int(np.asarray(a).nbytes) for a in self._data.values()) >= (1 << 20):
            return _write_in_process(self, tmp)
        return self._write_local(tmp)

    def _write_local(self, tmp):
        h = ctypes.c_int64(0)
        expected = sum(int(np.asarray(a).nbytes) for a in self._data.values()) + 65536
        self._check(self._lib.dio_create(tmp.encode(), expected, ctypes.byref(h)), "create %s" % tmp)
        try:
            for name, arr in self._data.items():
                arr = np.asarray(arr)
                if arr.dtype not in _NP2DIO:
                    arr = arr.astype(np.complex128 if arr.dtype.kind == "c" else (np.int64 if arr.dtype.kind in "iu" else np.float64))
                arr = np.ascontiguousarray(arr).reshape(arr.shape)   # ascontiguousarray turns 0-d into 1-d
                chunks, comp = self._opts.get(name, (None, None))
                if comp not in (None, "lzf", "bitshuffle"):
                    raise IOError("compression %r is not available (libdriftio writes lzf and bitshuffle + LZ4)" % (comp,))
                if comp is not None and chunks is None:
                    chunks = arr.shape   # h5py would guess a chunk shape; one chunk keeps small datasets simple
                if arr.ndim == 0 or arr.size == 0:
                    chunks, comp = None, None
                self._check(self._lib.dio_write_dataset(h, name.encode(), _NP2DIO[arr.dtype], arr.ndim, _u64(arr.shape),
                                                        _u64(chunks) if chunks is not None else None,
                                                        {"lzf": COMP_LZF, "bitshuffle": COMP_BSHUF_LZ4}.get(comp, COMP_NONE),
                                                        arr.ctypes.data_as(ctypes.c_void_p)), "write %s" % name)
            for k, v in self.attrs.items():
                self._write_attr(h, k, v)
        finally:
            self._lib.dio_close(h)

    def _write_attr(self, h, name, v):
        if isinstance(v, bytes):
            v = v.decode()
        if isinstance(v, str):
            self._check(self._lib.dio_write_attr(h, name.encode(), STR, 0, None, ctypes.c_char_p(v.encode())), "attr %s" % name)
            return
        a = np.asarray(v)
        if a.dtype.kind in "US":
            self._write_attr(h, name, str(a.item()) if a.shape == () else ",".join(str(x) for x in a.ravel()))
            return
        if a.dtype not in _NP2DIO:
            a = a.astype(np.complex128 if a.dtype.kind == "c" else (np.int64 if a.dtype.kind in "iu" else np.float64))
        a = np.ascontiguousarray(a).reshape(a.shape)
        self._check(self._lib.dio_write_attr(h, name.encode(), _NP2DIO[a.dtype], a.ndim, _u64(a.shape) if a.ndim else None,
                                             a.ctypes.data_as(ctypes.c_void_p)), "attr %s" % name)

    def _close_handle(self):
        if self._h is not None:
            self._lib.dio_close(self._h)
            self._h = None


class _H5pyTmp(object):
    """h5py.File opened for writing on a temporary name, renamed into place on close."""

    def __init__(self, path, **kwargs):
        self._path = path
        self._tmp = path + ".tmp%d_%d" % (os.getpid(), threading.get_ident())
        self._f = h5py.File(self._tmp, "w", **kwargs)

    def __getattr__(self, name):
        return getattr(self._f, name)

    def __getitem__(self, name):
        return self._f[name]

    def __contains__(self, name):
        return name in self._f

    def close(self):
        self._f.close()
        os.replace(self._tmp, self._path)

    def __enter__(self):
        return self

    def __exit__(self, et, ev, tb):
        if et is None:
            self.close()
        else:
            self._f.close()
            if os.path.exists(self._tmp):
                os.remove(self._tmp)
        return False


def File(path, mode="r", **kwargs):
    """Open a product file.  Existing files are opened by what they ARE (HDF5 or npz); new files are written
    with the configured back-end."""
    trace = os.environ.get("DRIFTMI_TRACE_OPEN")   # debugging / test aid: one "mode path" line per open
    if trace and (mode == "w" or os.path.exists(path)):
        with open(trace, "a") as fh:
            fh.write("%s %s\n" % (mode, path))
    if discard() and mode == "w":
        raise IOError("DRIFTMI_STORAGE=discard: %s would be written" % path)
    if mode in ("r", "r+", "a") and os.path.exists(path):
        if _is_hdf5(path):
            if HAVE_H5PY:
                return h5py.File(path, mode, **kwargs)
            if load_driftio() is None:
                raise IOError("%s is an HDF5 file and no HDF5 library is available" % path)
            return DriftioFile(path, mode)
        return NpzFile(path, mode)
    be = backend()
    if be == "h5py":
        return _H5pyTmp(path, **kwargs) if mode == "w" else h5py.File(path, mode, **kwargs)
    if be == "driftio":
        return DriftioFile(path, mode)
    return NpzFile(path, mode)


def trace_mark(text):
    """A marker line in the DRIFTMI_TRACE_OPEN file (test aid: "which files were opened before this point")."""
    trace = os.environ.get("DRIFTMI_TRACE_OPEN")
    if trace:
        with open(trace, "a") as fh:
            fh.write("# %s\n" % text)


def compression_kwargs(chunks):
    """create_dataset keywords for a chunked, compressed product dataset (drift/core/beamtransfer.py:548-555,
    :741-792): lzf, or bitshuffle + LZ4 (the reference's choice for truncated blocks) with DRIFTMI_H5_CODEC=bitshuffle.
    The npz mirror ignores them."""
    codec = "bitshuffle" if os.environ.get("DRIFTMI_H5_CODEC", "lzf") == "bitshuffle" else "lzf"
    return dict(chunks=tuple(int(c) for c in chunks), compression=codec)


# ---- background writers -------------------------------------------------------------------------
# Product files are independent per m: they are written by a small thread pool while the GPU works on
# the next batch (file output is 10x the compute time of BASELINE configs[1] when done inline).
# DRIFTMI_IO_THREADS = 0 writes inline.
_pool = None
_pending = []
_plock = threading.Lock()   # submit() may be called from several driver threads (bench --streams)


def discard():
    """``DRIFTMI_STORAGE=discard``: products are computed and left in HBM, no file is written (and none can be read
    back).  For measuring the compute part of a job on its own — the boundary of the hot path hands over device
    buffers; `bench.py --workload configs2` times the job both ways and reports the difference as file output."""
    return os.environ.get("DRIFTMI_STORAGE", "").lower() == "discard"


# ---- writer processes ------------------------------------------------------------------------------
# libhdf5 is not thread-safe: inside ONE process every HDF5 call sits behind a lock, and although the writer threads
# compress their chunks outside it, the chunk writes themselves (file-space allocation, chunk index, the copy into the
# page cache) run one at a time — 2 GB/s per rank, the larger part of a configs[2] job with files.  With
# DRIFTMI_IO_PROCS = N > 0 the writer threads keep their role (they run the closures that assemble a file) but hand
# the assembled file — dataset bytes through POSIX shared memory, names / chunk shapes / attributes pickled — to one of N
# worker PROCESSES, each with its own libhdf5, and wait for it: N files are compressed AND written at the same time.
# The workers are fresh interpreters (spawn: started as children, nothing of this process' GPU state is inherited); they
# import this module only.
_workers = None            # queue.Queue of idle _Worker objects
_worker_list = []
_proc_lock = threading.Lock()


class WriteFailed(IOError):
    """A writer process REPORTED a failed write (disk full, a bad dataset): the process itself is fine."""


class _Worker(object):
    """One writer process: `python -c "... io_worker.main()"` talking length-prefixed pickles over its pipes (not
    multiprocessing's spawn, which would re-import the caller's __main__ module in every child)."""

    def __init__(self):
        import subprocess
        import sys

        root = os.path.dirname(_HERE)
        code = "import sys; sys.path.insert(0, %r); from driftscan_amd import io_worker; io_worker.main()" % root
        env = dict(os.environ, DRIFTMI_IO_PROCS="0", OMP_NUM_THREADS="1")
        self.p = subprocess.Popen([sys.executable, "-c", code], stdin=subprocess.PIPE, stdout=subprocess.PIPE, env=env)

    def run(self, task):
        import pickle
        import struct

        blob = pickle.dumps(task, protocol=pickle.HIGHEST_PROTOCOL)
        self.p.stdin.write(struct.pack("<Q", len(blob)))
        self.p.stdin.write(blob)
        self.p.stdin.flush()
        hdr = self.p.stdout.read(8)
        if len(hdr) < 8:
            raise IOError("writer process died (exit code %s)" % self.p.poll())
        status, val = pickle.loads(self.p.stdout.read(struct.unpack("<Q", hdr)[0]))
        if status != "ok":
            raise WriteFailed("writer process: %s" % val)   # the process answered: it is alive and in step
        return val

    def stop(self):
        try:
            self.p.stdin.close()
            self.p.wait(timeout=30)
        except Exception:
            self.p.kill()


def _io_procs():
    if backend() != "driftio":
        return 0
    return int(os.environ.get("DRIFTMI_IO_PROCS", "0"))


def _proc_task(tmp, specs, attrs):
    """Worker side: attach the shared-memory blocks, write the file `tmp` with this process' libdriftio."""
    import mmap

    blocks = []
    try:
        f = DriftioFile.__new__(DriftioFile)
        f._lib = load_driftio()
        if f._lib is None:
            raise IOError("libdriftio is not available in the writer process")
        f.path, f.mode, f._h, f._infos = tmp, "w", None, {}
        f._data, f._opts = {}, {}
        f.attrs = dict(attrs)
        for name, shm_name, shape, dtype, chunks, comp in specs:
            if shm_name is None:
                f._data[name] = np.zeros(shape, dtype=np.dtype(dtype))
            else:
                # mapped by hand (POSIX shared memory lives under /dev/shm): multiprocessing.shared_memory would register the
                # block with THIS process' resource tracker, which then tries to unlink what the parent owns
                fd = os.open("/dev/shm/" + shm_name.lstrip("/"), os.O_RDWR)
                try:
                    blk = mmap.mmap(fd, 0)
                finally:
                    os.close(fd)
                blocks.append(blk)
                f._data[name] = np.ndarray(shape, dtype=np.dtype(dtype), buffer=blk)
            f._opts[name] = (chunks, comp)
        f._write_local(tmp)
        f._data.clear()
    finally:
        for blk in blocks:
            try:
                blk.close()
            except BufferError:
                pass
    return os.path.getsize(tmp)


def _write_in_process(f, tmp):
    """Writer-thread side: copy the datasets of `f` into shared memory, have a worker process write `tmp`, wait."""
    global _workers
    from multiprocessing import shared_memory

    with _proc_lock:
        if _workers is None:
            import queue

            _workers = queue.Queue()
            for _ in range(_io_procs()):
                w = _Worker()
                _worker_list.append(w)
                _workers.put(w)
        workers = _workers
    blocks, specs = [], []
    try:
        for name, arr in f._data.items():
            arr = np.asarray(arr)
            chunks, comp = f._opts.get(name, (None, None))
            if arr.size == 0:
                specs.append((name, None, arr.shape, arr.dtype.str, chunks, comp))
                continue
            blk = shared_memory.SharedMemory(create=True, size=int(arr.nbytes))
            blocks.append(blk)
            np.copyto(np.ndarray(arr.shape, dtype=arr.dtype, buffer=blk.buf), arr)
            specs.append((name, blk.name, arr.shape, arr.dtype.str, chunks, comp))
        w = workers.get()
        if w is None:
            # a slot whose process could not be restarted: the slot goes back as it is (the pool never shrinks, nobody
            # blocks in get() for ever) and this file is written here, in the writer thread
            workers.put(None)
            f._write_local(tmp)
            return
        try:
            w.run((tmp, specs, dict(f.attrs)))
        except WriteFailed:
            workers.put(w)      # a reported failure: the worker stays (respawning it would wait up to 30 s in stop())
            raise
        except BaseException:
            # transport failure (broken pipe, short read): the process may be gone or out of step — never hand it out
            # again; a fresh one takes its slot, or the slot stays as a None marker when none can be started
            try:
                w.p.kill()
            except Exception:
                pass
            fresh = None
            with _proc_lock:
                if w in _worker_list:
                    _worker_list.remove(w)
                try:
                    fresh = _Worker()
                    _worker_list.append(fresh)
                except Exception:
                    fresh = None
            workers.put(fresh)
            raise
        else:
            workers.put(w)
    finally:
        for blk in blocks:
            blk.close()
            blk.unlink()


def shutdown_writers():
    """Stop the writer processes (tests; a long-running driver keeps them)."""
    global _workers
    with _proc_lock:
        for w in _worker_list:
            w.stop()
        del _worker_list[:]
        _workers = None


_cap_bytes = None


def _pending_cap():
    """Bytes of products that may wait in the writer queue.  ``DRIFTMI_IO_PENDING_GB`` if set; otherwise a fifth of the
    host memory available at first use, shared between the ranks of the node, at most 32 GB: with 8 GB (rounds 1-3) a
    configs[2] rank had three files in flight and the GPU waited for the writers, 39.8 s per 25 m-blocks against 9.4 s of
    compute."""
    global _cap_bytes
    if _cap_bytes is None:
        e = os.environ.get("DRIFTMI_IO_PENDING_GB")
        if e:
            _cap_bytes = float(e) * (1 << 30)
        else:
            avail = 16 << 30
            try:
                with open("/proc/meminfo") as f:
                    for line in f:
                        if line.startswith("MemAvailable:"):
                            avail = int(line.split()[1]) * 1024
                            break
            except OSError:
                pass
            local = max(1, int(os.environ.get("LOCAL_WORLD_SIZE", "1")))
            _cap_bytes = float(min(32 << 30, max(2 << 30, avail // (5 * local))))
    return _cap_bytes


class Deferred:
    """A device tensor on its way into a file.  `Context.defer_host` records an event behind whatever produced it and
    hands it over; the COPY THREAD of the writer pool makes the host copy — page-locked allocation (a configs[2] share pins
    145 GB in its life, 8 GB/s) and the transfer, on a stream of its own behind that event — while the compute stream goes
    on with the next stage; the device block is let go as soon as its copy is complete.  Done inline (`Context.to_host`,
    rounds 1-3) the same copies were 8.3 of the 29 s of a share 0/16 with files, with the GPU idle.

    ``resident``: the caller keeps the tensor alive anyway (the beam transfer blocks of a range, which the SVD stage
    reads next) — it does not count against the bound on device memory held by the queue (``DRIFTMI_IO_DEVICE_GB``, 16)."""

    __slots__ = ("ctx", "t", "event", "nbytes", "resident")

    def __init__(self, ctx, t, event, resident=False):
        self.ctx, self.t, self.event, self.resident = ctx, t, event, bool(resident)
        self.nbytes = int(t.numel()) * int(t.element_size())

    def host(self):
        torch = self.ctx.torch
        t0 = time.perf_counter()
        with torch.cuda.device(self.ctx.device):
            h = torch.empty(self.t.shape, dtype=self.t.dtype, pin_memory=True)
            t1 = time.perf_counter()
            s = _copy_stream(self.ctx)
            s.wait_event(self.event)
            with torch.cuda.stream(s):
                h.copy_(self.t, non_blocking=True)
            s.synchronize()
        t2 = time.perf_counter()
        _io_stat("pinned_alloc_s", t1 - t0)
        _io_stat("copy_wait_and_transfer_s", t2 - t1)
        _io_stat("copied_bytes", self.nbytes)
        self.t = self.event = None
        return h.numpy()


_copy_streams = {}
_copier = None
# where the writer pipeline spends its time (seconds summed over its threads; `io_stats()`): the copy thread's page-locked
# allocations and transfers, its waits for room on the host, the writer threads' closures (gather + compress + HDF5)
_io_stats = {}
_io_stats_lock = threading.Lock()


def _io_stat(key, val):
    with _io_stats_lock:
        _io_stats[key] = _io_stats.get(key, 0.0) + val


def io_stats(reset=False):
    with _io_stats_lock:
        out = dict(_io_stats)
        if reset:
            _io_stats.clear()
    return out


# what the queue holds: host bytes (products copied or being copied, until their file is written), tasks, and device
# bytes (deferred products not yet copied, other than `resident` ones)
_cv = threading.Condition()
_held = {"host": 0, "count": 0, "dev": 0}


def _writer_init():
    """Writer threads (and the chunk compressors they start, which inherit it) run at a lower priority than the thread
    that drives the GPU: with more of them than the rank has cores, kernel launches would otherwise queue behind LZ4."""
    try:
        os.setpriority(os.PRIO_PROCESS, threading.get_native_id(), int(os.environ.get("DRIFTMI_IO_NICE", "10")))
    except (OSError, AttributeError):
        pass


def _copy_stream(ctx):
    if ctx.device not in _copy_streams:
        _copy_streams[ctx.device] = ctx.torch.cuda.Stream(device=ctx.device)
    return _copy_streams[ctx.device]


def _device_cap():
    return float(os.environ.get("DRIFTMI_IO_DEVICE_GB", "16")) * (1 << 30)


def _host_acquire(nbytes, nthreads):
    """Blocks until the queue has room for another task of ``nbytes`` (always admits one when it is empty)."""
    cap = _pending_cap()
    small = nbytes < (32 << 20)   # a KL or marker file does not wait behind the gigabytes of beam and SVD products
    with _cv:
        while _held["count"] > 0 and (_held["count"] >= 64 * nthreads or (not small and _held["host"] + nbytes > cap)):
            _cv.wait()
        _held["host"] += nbytes
        _held["count"] += 1


def _release(host=0, dev=0, count=0):
    with _cv:
        _held["host"] -= host
        _held["dev"] -= dev
        _held["count"] -= count
        _cv.notify_all()


class _Chained:
    """Future of a task that goes through the copy thread first: the copy, then the write it queued."""

    def __init__(self, outer):
        self.outer = outer

    def result(self):
        return self.outer.result().result()

    def done(self):
        return self.outer.done() and (self.outer.exception() is not None or self.outer.result().done())

    def exception(self):
        return self.outer.exception() or self.outer.result().exception()


def _resolve(args):
    return tuple(a.host() if isinstance(a, Deferred) else a for a in args)


def submit(fn, *args):
    """Run ``fn(*args)`` (a closure that writes one file) on the writer pool.  `Deferred` arguments reach ``fn`` as numpy
    arrays: the copy thread makes them, in submission order, when the queue has room for them on the host, and then queues
    the write — the caller only waits if the device memory held that way passes ``DRIFTMI_IO_DEVICE_GB``.  Plain numpy
    arguments wait for room here."""
    global _pool, _copier
    if discard():
        return
    nthreads = int(os.environ.get("DRIFTMI_IO_THREADS", "8"))
    if nthreads <= 0:
        fn(*_resolve(args))
        return
    # bound the host memory held by queued products: by count and by bytes (a configs[2] m-block is 1.8 GB of beam_m
    # and 2.6 GB of SVD products)
    nbytes = sum(int(a.nbytes) for a in args if isinstance(a, (np.ndarray, Deferred)))
    deferred = [a for a in args if isinstance(a, Deferred)]
    with _plock:
        if _pool is None:
            from concurrent.futures import ThreadPoolExecutor

            _pool = ThreadPoolExecutor(max_workers=nthreads, thread_name_prefix="driftmi-io", initializer=_writer_init)
            _copier = ThreadPoolExecutor(max_workers=1, thread_name_prefix="driftmi-copy")
        pool, copier = _pool, _copier

    def write():
        t0 = time.perf_counter()
        try:
            return fn(*write.args)
        finally:
            write.args = None
            _release(host=nbytes, count=1)
            _io_stat("writer_closure_s", time.perf_counter() - t0)
            _io_stat("files", 1)

    if deferred:
        dev = sum(a.nbytes for a in deferred if not a.resident)
        if dev:
            cap = _device_cap()
            with _cv:
                while _held["dev"] > 0 and _held["dev"] + dev > cap:
                    _cv.wait()
                _held["dev"] += dev

        def copy():
            got = False
            try:
                t0 = time.perf_counter()
                _host_acquire(nbytes, nthreads)
                _io_stat("copy_thread_wait_for_host_room_s", time.perf_counter() - t0)
                got = True
                write.args = _resolve(args)
            except BaseException:
                if got:
                    _release(host=nbytes, count=1)
                raise
            finally:
                _release(dev=dev)
            return pool.submit(write)

        fut = _Chained(copier.submit(copy))
    else:
        _host_acquire(nbytes, nthreads)
        write.args = args
        fut = pool.submit(write)
    with _plock:
        while _pending and _pending[0].done() and _pending[0].exception() is None:
            _pending.pop(0)
        _pending.append(fut)


def wait_copies():
    """Wait until every `Deferred` handed to `submit` so far has been copied to the host (NOT until its file is written):
    the copy thread works in submission order, so a marker task behind them says when.  `BeamTransfer.generate` calls this
    before it allocates the beam blocks of its next range of m: the views held by the queue keep the previous range's
    allocation (up to `beam_chunk_gb`) alive, and those bytes are outside the `DRIFTMI_IO_DEVICE_GB` bound."""
    with _plock:
        copier = _copier
    if copier is not None:
        copier.submit(lambda: None).result()


def flush():
    """Wait for every queued write (re-raises the first failure).  Called before anything reads the files."""
    while True:
        with _plock:
            if not _pending:
                return
            w = _pending.pop(0)
        w.result()


def can_open(path):
    try:
        f = File(path, "r")
        f.close()
        return True
    except Exception:
        return False
