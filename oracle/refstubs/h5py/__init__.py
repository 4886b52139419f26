"""In-memory look-alike for the small part of the h5py API the reference uses.

Files live in a process-global dict keyed by normalised path, so a file written
by one `File(path, "w")` can be re-opened by a later `File(path, "r")`.
TEST INFRASTRUCTURE ONLY (see README.md in this directory).
"""
import os

import numpy as np

_STORE = {}


class Dataset(object):
    def __init__(self, arr):
        self._arr = arr
        self.attrs = {}

    @property
    def shape(self):
        return self._arr.shape

    @property
    def dtype(self):
        return self._arr.dtype

    def __getitem__(self, ind):
        out = self._arr[ind]
        return np.array(out) if isinstance(out, np.ndarray) else out

    def __setitem__(self, ind, val):
        self._arr[ind] = val

    def __len__(self):
        return self._arr.shape[0]


class File(object):
    def __init__(self, path, mode="r", **kwargs):
        key = os.path.normpath(str(path))
        if mode in ("w", "w-", "x"):
            _STORE[key] = {"dsets": {}, "attrs": {}}
        elif mode in ("r", "r+"):
            if key not in _STORE:
                raise IOError("no such (in-memory) file: %s" % key)
        elif mode == "a":
            _STORE.setdefault(key, {"dsets": {}, "attrs": {}})
        self._f = _STORE[key]
        self.attrs = self._f["attrs"]
        self.filename = key

    def create_dataset(self, name, shape=None, dtype=None, data=None, **kwargs):
        if data is not None:
            arr = np.array(data, dtype=dtype) if dtype is not None else np.array(data)
        else:
            arr = np.zeros(shape, dtype=dtype)
        d = Dataset(arr)
        self._f["dsets"][name.lstrip("/")] = d   # "/mmode" and "mmode" name the same dataset of the root group
        return d

    def __getitem__(self, name):
        return self._f["dsets"][name.lstrip("/")]

    def __contains__(self, name):
        return name.lstrip("/") in self._f["dsets"]

    def keys(self):
        return self._f["dsets"].keys()

    def close(self):
        pass

    def __enter__(self):
        return self

    def __exit__(self, *a):
        return False


def exists(path):
    return os.path.normpath(str(path)) in _STORE
