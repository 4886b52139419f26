"""Product files: real HDF5 through h5py when it is importable, otherwise a
directory-free single-file ``.npz`` mirror with the same group/dataset/attribute
names.  The reference's on-disk layout is preserved by name: the
same paths (``bt/beam_m/<m>/beam.hdf5`` ...), dataset names and attributes; with
the mirror back-end the file at that path holds an npz archive instead of HDF5
(h5py is not part of this image's primary Python).
"""
import os

import numpy as np

try:  # pragma: no cover - depends on the environment
    import h5py  # noqa: F401

    HAVE_H5PY = True
except Exception:  # pragma: no cover
    h5py = None
    HAVE_H5PY = False


class _NpzDataset(object):
    def __init__(self, owner, name):
        self._o, self._n = owner, name

    def _arr(self):
        return self._o._load(self._n)

    @property
    def shape(self):
        return self._arr().shape

    @property
    def dtype(self):
        return self._arr().dtype

    def __getitem__(self, idx):
        out = self._arr()[idx]
        return np.array(out) if isinstance(out, np.ndarray) else out

    def __setitem__(self, idx, val):
        self._arr()[idx] = val
        self._o._dirty = True

    def __len__(self):
        return self.shape[0]


class NpzFile(object):
    """Minimal h5py.File look-alike persisted as one ``.npz`` archive at ``path``."""

    def __init__(self, path, mode="r"):
        self.path, self.mode = path, mode
        self._data, self.attrs, self._dirty = {}, {}, False
        self._lazy = None   # open archive of a file being read: datasets are loaded when first touched
        if mode in ("r", "r+", "a") and os.path.exists(path):
            z = np.load(path, allow_pickle=False)
            for k in z.files:
                if k.startswith("__attr__"):
                    v = z[k]
                    self.attrs[k[8:]] = v.item() if v.shape == () else v
                else:
                    self._data[k] = None
            self._lazy = z
        elif mode == "r":
            raise IOError("no such file: %s" % path)
        if mode == "w":
            self._dirty = True

    def create_dataset(self, name, shape=None, dtype=None, data=None, **kwargs):
        if data is not None:
            # no copy for arrays handed over for writing (the products are hundreds of MB per file)
            arr = np.asarray(data, dtype=dtype) if dtype is not None else np.asarray(data)
        else:
            arr = np.zeros(shape, dtype=dtype)
        self._data[name] = arr
        self._dirty = True
        return _NpzDataset(self, name)

    def _load(self, name):
        if self._data[name] is None:
            self._data[name] = self._lazy[name]
        return self._data[name]

    def __getitem__(self, name):
        if name not in self._data:
            raise KeyError(name)
        return _NpzDataset(self, name)

    def __contains__(self, name):
        return name in self._data

    def keys(self):
        return self._data.keys()

    def close(self):
        if self.mode != "r" and (self._dirty or self.attrs):
            payload = {k: self._load(k) for k in self._data}
            for k, v in self.attrs.items():
                payload["__attr__" + k] = np.asarray(v)
            tmp = self.path + ".tmp%d" % os.getpid()
            with open(tmp, "wb") as fh:
                np.savez(fh, **payload)
            os.replace(tmp, self.path)  # write-temp-then-rename, like caput.misc.lock_file
            self._dirty = False
        if self._lazy is not None:
            self._lazy.close()
            self._lazy = None

    def __enter__(self):
        return self

    def __exit__(self, *exc):
        self.close()
        return False


def File(path, mode="r", **kwargs):
    """Open a product file with the best available back-end."""
    if HAVE_H5PY:
        return h5py.File(path, mode, **kwargs)
    return NpzFile(path, mode)


# ---- background writers -------------------------------------------------------------------------
# Product files are independent per m: they are written by a small thread pool while the GPU works on
# the next batch (file output is 10x the compute time of BASELINE configs[1] when done inline).
# DRIFTMI_IO_THREADS = 0 writes inline.
import threading

_pool = None
_pending = []
_plock = threading.Lock()   # submit() may be called from several driver threads (bench --streams)


def submit(fn, *args):
    """Run ``fn(*args)`` (a closure that writes one file) on the writer pool."""
    global _pool
    nthreads = int(os.environ.get("DRIFTMI_IO_THREADS", "8"))
    if nthreads <= 0:
        fn(*args)
        return
    with _plock:
        if _pool is None:
            from concurrent.futures import ThreadPoolExecutor

            _pool = ThreadPoolExecutor(max_workers=nthreads, thread_name_prefix="driftmi-io")
        wait = []
        while len(_pending) >= 4 * nthreads:   # bound the host memory held by queued products
            wait.append(_pending.pop(0))
        _pending.append(_pool.submit(fn, *args))
    for w in wait:
        w.result()


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
