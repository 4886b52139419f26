"""ctypes binding of libdriftcomm.so (include/driftcomm.h): gather and all-reduce of doubles over RCCL for hosts
that do not use torch.distributed.

BINDING ONLY — deliberately not a second collective stack of this package.  The pipeline classes talk through
``parallel.py`` (torch.distributed, backend "nccl" = RCCL) and never load this module; ``parallel.allreduce_sum`` and the
gathers do NOT fall back to it.  It exists for a driftscan maintainer whose launcher is MPI without torch: the C ABI alone
(libdriftmi + libdriftcomm) is then enough to run the m-sharded job on several GPUs, one communicator per process = one
rank per GPU, the unique id travelling over the host's own channel.  Exercised by
tests/test_gpu_primitives.py (one rank: the pool gives one GPU per call) and tests/test_cabi_symbols.py (exports)."""
import ctypes
import os

_HERE = os.path.dirname(os.path.abspath(__file__))
LIBPATH = os.path.join(_HERE, "lib", "libdriftcomm.so")

c_vp, c_int, c_sz = ctypes.c_void_p, ctypes.c_int, ctypes.c_size_t
SIGNATURES = {
    "dm_comm_last_error": (ctypes.c_char_p, []),
    "dm_comm_unique_id": (c_int, [c_vp]),
    "dm_comm_init_rank": (c_int, [c_int, c_int, c_vp, c_int, c_vp, ctypes.POINTER(c_vp)]),
    "dm_comm_destroy": (c_int, [c_vp]),
    "dm_comm_rank": (c_int, [c_vp]),
    "dm_comm_size": (c_int, [c_vp]),
    "dm_allreduce_f64": (c_int, [c_vp, c_vp, c_sz, c_vp]),
    "dm_gather_f64": (c_int, [c_vp, c_vp, c_vp, c_sz, c_int, c_vp]),
    "dm_comm_sync": (c_int, [c_vp]),
}
ID_BYTES = 128
_lib = None


def load():
    global _lib
    if _lib is None:
        lib = ctypes.CDLL(LIBPATH)
        for name, (res, args) in SIGNATURES.items():
            fn = getattr(lib, name)
            fn.restype, fn.argtypes = res, args
        _lib = lib
    return _lib


def unique_id():
    buf = ctypes.create_string_buffer(ID_BYTES)
    lib = load()
    if lib.dm_comm_unique_id(buf) != 0:
        raise RuntimeError(lib.dm_comm_last_error().decode())
    return buf.raw


class Communicator(object):
    def __init__(self, nranks, rank, uid, device=0, stream=None):
        self.lib = load()
        h = c_vp()
        idbuf = ctypes.create_string_buffer(uid, ID_BYTES)
        rc = self.lib.dm_comm_init_rank(int(nranks), int(rank), idbuf, int(device), c_vp(stream) if stream else None,
                                        ctypes.byref(h))
        if rc != 0:
            raise RuntimeError(self.lib.dm_comm_last_error().decode())
        self.h = h

    def _check(self, rc):
        if rc != 0:
            raise RuntimeError(self.lib.dm_comm_last_error().decode())

    @property
    def rank(self):
        return self.lib.dm_comm_rank(self.h)

    @property
    def size(self):
        return self.lib.dm_comm_size(self.h)

    @staticmethod
    def _user_stream(stream):
        """The stream the buffers live on: the caller's, or torch's current stream (the collective is ordered after the
        work enqueued there, and later work there after the collective — include/driftcomm.h)."""
        if stream is None:
            import torch

            stream = torch.cuda.current_stream().cuda_stream
        return c_vp(stream) if stream else None

    def allreduce(self, t, stream=None):
        """In-place sum of a float64 device tensor over the ranks."""
        self._check(self.lib.dm_allreduce_f64(self.h, c_vp(t.data_ptr()), int(t.numel()), self._user_stream(stream)))

    def gather(self, send, recv, root=0, stream=None):
        self._check(self.lib.dm_gather_f64(self.h, c_vp(send.data_ptr()), c_vp(recv.data_ptr()) if recv is not None else None,
                                           int(send.numel()), int(root), self._user_stream(stream)))

    def sync(self):
        self._check(self.lib.dm_comm_sync(self.h))

    def close(self):
        if getattr(self, "h", None):
            self.lib.dm_comm_destroy(self.h)
            self.h = None
