"""Product files: real HDF5 through h5py when it is importable, otherwise a
directory-free single-file ``.npz`` mirror with the same group/dataset/attribute
names.  The reference's on-disk layout is preserved by name: the
same paths (``bt/beam_m/<m>/beam.hdf5`` ...), dataset names and attributes; with
the mirror back-end the file at that path holds an npz archive instead of HDF5
(h5py is not part of this image's primary Python).
"""
import io
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

    @property
    def shape(self):
        return self._o._data[self._n].shape

    @property
    def dtype(self):
        return self._o._data[self._n].dtype

    def __getitem__(self, idx):
        out = self._o._data[self._n][idx]
        return np.array(out) if isinstance(out, np.ndarray) else out

    def __setitem__(self, idx, val):
        self._o._data[self._n][idx] = val
        self._o._dirty = True

    def __len__(self):
        return self.shape[0]


class NpzFile(object):
    """Minimal h5py.File look-alike persisted as one ``.npz`` archive at ``path``."""

    def __init__(self, path, mode="r"):
        self.path, self.mode = path, mode
        self._data, self.attrs, self._dirty = {}, {}, False
        if mode in ("r", "r+", "a") and os.path.exists(path):
            with np.load(path, allow_pickle=False) as z:
                for k in z.files:
                    if k.startswith("__attr__"):
                        v = z[k]
                        self.attrs[k[8:]] = v.item() if v.shape == () else v
                    else:
                        self._data[k] = z[k]
        elif mode == "r":
            raise IOError("no such file: %s" % path)
        if mode == "w":
            self._dirty = True

    def create_dataset(self, name, shape=None, dtype=None, data=None, **kwargs):
        if data is not None:
            arr = np.array(data, dtype=dtype) if dtype is not None else np.array(data)
        else:
            arr = np.zeros(shape, dtype=dtype)
        self._data[name] = arr
        self._dirty = True
        return _NpzDataset(self, name)

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
            payload = dict(self._data)
            for k, v in self.attrs.items():
                payload["__attr__" + k] = np.asarray(v)
            buf = io.BytesIO()
            np.savez(buf, **payload)
            tmp = self.path + ".tmp%d" % os.getpid()
            with open(tmp, "wb") as fh:
                fh.write(buf.getvalue())
            os.replace(tmp, self.path)  # write-temp-then-rename, like caput.misc.lock_file
            self._dirty = False

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


def can_open(path):
    try:
        f = File(path, "r")
        f.close()
        return True
    except Exception:
        return False
