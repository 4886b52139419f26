"""ctypes binding of libdriftmi.so (the C ABI declared in include/driftmi.h).

The library is built in-tree by ``__graft_entry__.build()`` / ``make -C
driftscan_amd/csrc`` into ``driftscan_amd/lib/libdriftmi.so``.  Loading fails
loudly if it is missing; there is no alternative code path.
"""
import ctypes
import os

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
LIBPATH = os.path.join(_HERE, "lib", "libdriftmi.so")
if os.environ.get("DRIFTMI_LIB"):  # a differently built libdriftmi.so (kernel tuning experiments)
    LIBPATH = os.environ["DRIFTMI_LIB"]

c_int = ctypes.c_int
c_i64 = ctypes.c_int64
c_dbl = ctypes.c_double
c_vp = ctypes.c_void_p
c_sz = ctypes.c_size_t

class ZgemmProblem(ctypes.Structure):
    """dm_zgemm_problem of include/driftmi.h."""
    _fields_ = [("A", c_vp), ("B", c_vp), ("C", c_vp), ("M", c_int), ("N", c_int), ("K", c_int),
                ("rsA", c_int), ("csA", c_int), ("rsB", c_int), ("csB", c_int), ("ldc", c_int),
                ("conjA", c_int), ("conjB", c_int), ("alpha", c_dbl), ("beta", c_dbl)]


# name -> (restype, argtypes); mirrors include/driftmi.h one to one
SIGNATURES = {
    "dm_ctx_create": (c_int, [c_int, c_sz, c_vp, ctypes.POINTER(c_vp)]),
    "dm_ctx_destroy": (c_int, [c_vp]),
    "dm_ctx_sync": (c_int, [c_vp]),
    "dm_ctx_workspace_bytes": (c_sz, [c_vp]),
    "dm_ctx_workspace_reset": (c_int, [c_vp, c_sz]),
    "dm_last_error": (ctypes.c_char_p, [c_vp]),
    "dm_version": (c_int, []),
    "dm_prof_reset": (c_int, [c_vp, c_int]),
    "dm_prof_trd_stride": (c_int, []),
    "dm_prof_report": (c_int, [c_vp, ctypes.POINTER(c_dbl), ctypes.POINTER(c_dbl), ctypes.POINTER(ctypes.c_longlong)]),
    "dm_zgemm_strided_batched": (
        c_int,
        [c_vp, c_int, c_int, c_int, c_dbl, c_vp, c_int, c_int, c_int, c_i64, c_vp, c_int, c_int, c_int, c_i64,
         c_dbl, c_vp, c_int, c_i64, c_vp, c_i64, c_int],
    ),
    "dm_zgemm_grouped": (c_int, [c_vp, c_int, ctypes.POINTER(ZgemmProblem)]),
    "dm_zpotrf_batched": (c_int, [c_vp, c_int, c_vp, c_int, c_i64, c_int, ctypes.POINTER(c_int)]),
    "dm_ztrsm_left_lower_batched": (
        c_int, [c_vp, c_int, c_int, c_vp, c_int, c_i64, c_vp, c_int, c_i64, c_int, c_int]),
    "dm_jacobi_rows_batched": (
        c_int, [c_vp, c_int, c_int, c_int, c_int, c_vp, c_int, c_i64, c_int, c_vp, ctypes.POINTER(c_int)]),
    "dm_jacobi_herm_batched": (
        c_int, [c_vp, c_int, c_vp, c_int, c_i64, c_vp, c_int, c_i64, c_int, c_vp, ctypes.POINTER(c_int)]),
    "dm_herm_eig_batched": (c_int, [c_vp, c_int, c_vp, c_int, c_i64, c_vp, c_int, c_i64, c_int, c_vp]),
    "dm_svd_chain": (
        c_int, [c_vp, c_int, c_int, c_int, c_int, c_int, c_vp, c_vp, c_dbl, c_vp, c_vp, c_vp, c_vp,
                ctypes.POINTER(c_int), ctypes.POINTER(c_int)]),
    "dm_svd_chain_lmin": (
        c_int, [c_vp, c_int, c_int, c_int, c_int, c_int, ctypes.POINTER(c_int), c_vp, c_vp, c_dbl, c_vp, c_vp, c_vp, c_vp,
                ctypes.POINTER(c_int), ctypes.POINTER(c_int)]),
    "dm_project_cov": (
        c_int, [c_vp, c_int, c_int, c_int, c_int, c_int, c_vp, ctypes.POINTER(c_int), ctypes.POINTER(c_int), c_vp,
                c_int, ctypes.POINTER(c_int), c_vp, ctypes.POINTER(c_i64), c_int]),
    "dm_project_diag": (
        c_int, [c_vp, c_int, c_int, c_int, c_int, c_vp, ctypes.POINTER(c_int), c_vp, c_dbl, c_vp,
                ctypes.POINTER(c_i64), c_int]),
    "dm_regularise": (c_int, [c_vp, c_int, ctypes.POINTER(c_int), c_vp, ctypes.POINTER(c_i64), c_dbl]),
    "dm_eigh_gen": (
        c_int, [c_vp, c_int, ctypes.POINTER(c_int), c_vp, c_vp, ctypes.POINTER(c_i64), c_vp, ctypes.POINTER(c_i64),
                c_vp, ctypes.POINTER(c_dbl), ctypes.POINTER(c_int), c_int, c_dbl, ctypes.POINTER(c_int)]),
    "dm_kl_m": (
        c_int, [c_vp, c_int, c_int, c_int, c_int, c_int, c_int, c_vp, c_vp, ctypes.POINTER(c_int), ctypes.POINTER(c_int), c_vp,
                ctypes.POINTER(c_int), c_int, c_vp, ctypes.POINTER(c_int), c_int, c_vp, c_dbl, c_dbl, c_int, c_dbl, c_vp,
                ctypes.POINTER(c_i64), c_vp, ctypes.POINTER(c_i64), ctypes.POINTER(c_dbl), ctypes.POINTER(c_int)]),
    "dm_doublekl_m": (
        c_int, [c_vp, c_int, c_int, c_int, c_int, c_int, c_int, c_vp, c_vp, ctypes.POINTER(c_int), ctypes.POINTER(c_int), c_vp,
                ctypes.POINTER(c_int), c_int, c_vp, ctypes.POINTER(c_int), c_int, c_vp, c_dbl, c_dbl, c_dbl, c_int, c_dbl, c_vp,
                c_vp, ctypes.POINTER(c_i64), c_vp, ctypes.POINTER(c_i64), ctypes.POINTER(c_int), ctypes.POINTER(c_int),
                ctypes.POINTER(c_dbl)]),
    "dm_fisher": (
        c_int, [c_vp, c_int, c_int, c_int, c_int, c_int, c_vp, ctypes.POINTER(c_int), ctypes.POINTER(c_int), c_int,
                c_vp, c_vp, ctypes.POINTER(c_i64), ctypes.POINTER(c_int), c_vp, ctypes.POINTER(c_i64), c_vp, c_int]),
    "dm_bt_beam_cyl": (
        c_int, [c_vp, c_int, ctypes.POINTER(c_dbl), ctypes.POINTER(c_dbl), ctypes.POINTER(c_dbl), c_int,
                ctypes.POINTER(c_dbl), ctypes.POINTER(c_dbl), ctypes.POINTER(c_dbl), c_int, c_dbl, c_vp]),
    "dm_bt_beams_cyl": (
        c_int, [c_vp, c_int, ctypes.POINTER(c_dbl), ctypes.POINTER(c_dbl), ctypes.POINTER(c_dbl), c_int,
                ctypes.POINTER(c_int), ctypes.POINTER(c_dbl), ctypes.POINTER(c_dbl), ctypes.POINTER(c_dbl),
                ctypes.POINTER(c_int), ctypes.POINTER(c_dbl), c_vp, ctypes.c_size_t]),
    "dm_bt_maps": (
        c_int, [c_vp, c_int, ctypes.POINTER(c_dbl), ctypes.POINTER(c_dbl), ctypes.POINTER(c_dbl), c_int, c_int, c_vp,
                c_int, ctypes.POINTER(c_dbl), ctypes.POINTER(c_int), ctypes.POINTER(c_int), c_vp]),
    "dm_bt_maps_c": (
        c_int, [c_vp, c_int, ctypes.POINTER(c_dbl), ctypes.POINTER(c_dbl), ctypes.POINTER(c_dbl), c_int, c_int, c_vp,
                c_int, ctypes.POINTER(c_dbl), ctypes.POINTER(c_int), ctypes.POINTER(c_int), c_vp]),
    "dm_bt_sht": (
        c_int, [c_vp, c_int, ctypes.POINTER(c_dbl), ctypes.POINTER(c_dbl), c_int, c_int, c_int, c_int, c_int, c_int,
                c_int, ctypes.POINTER(c_int), ctypes.POINTER(c_int), ctypes.POINTER(c_int), c_vp, c_vp]),
    "dm_bt_sht_range": (
        c_int, [c_vp, c_int, ctypes.POINTER(c_dbl), ctypes.POINTER(c_dbl), c_int, c_int, c_int, c_int, c_int, c_int,
                c_int, c_int, ctypes.POINTER(c_int), ctypes.POINTER(c_int), ctypes.POINTER(c_int), c_vp, c_vp]),
    "dm_bt_sht_opts": (
        c_int, [c_vp, c_int, ctypes.POINTER(c_dbl), ctypes.POINTER(c_dbl), c_int, c_int, c_int, c_int, c_int, c_int,
                c_int, c_int, ctypes.POINTER(c_int), ctypes.POINTER(c_int), ctypes.POINTER(c_int), c_vp, c_vp, c_int,
                ctypes.POINTER(c_dbl)]),
    "dm_bt_columns": (
        c_int, [c_vp, c_int, ctypes.POINTER(c_dbl), ctypes.POINTER(c_dbl), ctypes.POINTER(c_dbl), c_int, c_int, c_vp, c_int,
                ctypes.POINTER(c_dbl), ctypes.POINTER(c_int), ctypes.POINTER(c_int), c_int, c_int, c_int, c_int, c_int, c_int,
                ctypes.POINTER(c_int), ctypes.POINTER(c_int), ctypes.POINTER(c_int), c_vp, ctypes.POINTER(c_dbl)]),
    "dm_bt_columns_c": (
        c_int, [c_vp, c_int, ctypes.POINTER(c_dbl), ctypes.POINTER(c_dbl), ctypes.POINTER(c_dbl), c_int, c_int, c_vp, c_int,
                ctypes.POINTER(c_dbl), ctypes.POINTER(c_int), ctypes.POINTER(c_int), c_int, c_int, c_int, c_int, c_int, c_int,
                ctypes.POINTER(c_int), ctypes.POINTER(c_int), ctypes.POINTER(c_int), c_vp, ctypes.POINTER(c_dbl)]),
    "dm_bt_columns_iter": (
        c_int, [c_vp, c_int, ctypes.POINTER(c_dbl), ctypes.POINTER(c_dbl), ctypes.POINTER(c_dbl), c_int, c_int, c_vp, c_int, c_int,
                ctypes.POINTER(c_dbl), ctypes.POINTER(c_int), ctypes.POINTER(c_int), c_int, c_int, c_int, c_int, c_int, c_int,
                ctypes.POINTER(c_int), ctypes.POINTER(c_int), ctypes.POINTER(c_int), c_vp, ctypes.POINTER(c_dbl), c_int]),
    "dm_bt_alias_info": (
        c_int, [c_int, ctypes.POINTER(c_dbl), ctypes.POINTER(c_dbl), c_int, c_int, ctypes.POINTER(c_int),
                ctypes.POINTER(c_int)]),
    "dm_bit_truncate_max_complex": (c_int, [c_vp, c_vp, c_i64, c_int, c_i64, c_dbl, c_dbl]),
}

_lib = None


class DriftMIError(RuntimeError):
    pass


def load():
    """Load libdriftmi.so and declare every prototype.  Raises if absent."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIBPATH):
        raise DriftMIError(
            "libdriftmi.so not found at %s — build it with `python -c 'import __graft_entry__ as g; g.build()'` "
            "(there is no CPU fallback)" % LIBPATH
        )
    lib = ctypes.CDLL(LIBPATH)
    for name, (res, args) in SIGNATURES.items():
        fn = getattr(lib, name)  # AttributeError if the symbol is missing: loud by design
        fn.restype = res
        fn.argtypes = args
    _lib = lib
    return lib


class Context(object):
    """One per GPU / process.  Binds to torch's current stream on that device so
    that torch.cuda events bracket the work."""

    def __init__(self, device=0, workspace_bytes=1 << 30):
        import torch

        if not torch.cuda.is_available():
            raise DriftMIError("no GPU visible: libdriftmi has no CPU fallback")
        self.torch = torch
        self.lib = load()
        self.device = int(device)
        torch.cuda.set_device(self.device)
        stream = torch.cuda.current_stream(self.device).cuda_stream
        h = c_vp()
        rc = self.lib.dm_ctx_create(self.device, int(workspace_bytes), c_vp(stream), ctypes.byref(h))
        if rc != 0:
            raise DriftMIError("dm_ctx_create failed with code %d" % rc)
        self.h = h

    def close(self):
        if getattr(self, "h", None):
            self.lib.dm_ctx_destroy(self.h)
            self.h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def check(self, rc, what=""):
        """Raise on every non-zero status of the C ABI: < 0 argument/runtime errors, > 0 the LAPACK-`info`-like numerical
        failures (an eigen-iteration that did not converge, a B that stays indefinite) — none of the entry points hands
        back usable products with one (a chain that went on after a failed Gram eigenproblem would be silently wrong)."""
        if rc != 0:
            msg = self.lib.dm_last_error(self.h)
            kind = "failed" if rc < 0 else "numerical failure, info ="
            raise DriftMIError("%s %s (%d): %s" % (what, kind, rc, msg.decode() if msg else ""))
        return rc

    def sync(self):
        self.check(self.lib.dm_ctx_sync(self.h), "dm_ctx_sync")

    # ---- device array helpers (torch is used for memory only) -------------
    def to_device(self, arr):
        t = self.torch.from_numpy(np.ascontiguousarray(arr))
        return t.to("cuda:%d" % self.device)

    def workspace_reset(self, nbytes=0):
        """Give the (idle) workspace arena back to the driver and optionally reserve ``nbytes`` afresh."""
        self.check(self.lib.dm_ctx_workspace_reset(self.h, int(nbytes)), "dm_ctx_workspace_reset")

    def to_host(self, t):
        """Device tensor -> numpy through page-locked memory (torch's caching host allocator re-uses the
        blocks): several times the rate of a pageable copy, which is what bounds the file output."""
        h = self.torch.empty(t.shape, dtype=t.dtype, pin_memory=True)
        h.copy_(t)
        return h.numpy()

    def defer_host(self, t, event=None, resident=False):
        """Device tensor -> `storage.Deferred`: the host copy is made by the writer pool's copy thread, behind an event
        recorded here (or ``event``, for several views of one result), so the caller goes straight on.  The tensor must
        not be written to afterwards."""
        from . import storage

        if event is None:
            event = self.record_event()
        return storage.Deferred(self, t, event, resident=resident)

    def record_event(self):
        ev = self.torch.cuda.Event()
        ev.record(self.torch.cuda.current_stream(self.device))
        return ev

    def empty(self, shape, dtype):
        tdt = {np.dtype(np.complex128): self.torch.complex128, np.dtype(np.float64): self.torch.float64,
               np.dtype(np.int32): self.torch.int32, np.dtype(np.int64): self.torch.int64}[np.dtype(dtype)]
        return self.torch.empty(shape, dtype=tdt, device="cuda:%d" % self.device)

    def zeros(self, shape, dtype):
        t = self.empty(shape, dtype)
        t.zero_()
        return t

    @staticmethod
    def ptr(t):
        return c_vp(t.data_ptr()) if t is not None else c_vp(0)

    # ---- thin wrappers over the dense building blocks ---------------------
    def zgemm(self, A, B, C, M, N, K, rsA, csA, rsB, csB, ldc, conjA=False, conjB=False, alpha=1.0, beta=0.0,
              kscale=None, batch=1, strideA=0, strideB=0, strideC=0, stride_kscale=0):
        rc = self.lib.dm_zgemm_strided_batched(
            self.h, M, N, K, alpha, self.ptr(A), rsA, csA, int(conjA), strideA, self.ptr(B), rsB, csB, int(conjB),
            strideB, beta, self.ptr(C), ldc, strideC, self.ptr(kscale), stride_kscale, batch)
        self.check(rc, "dm_zgemm_strided_batched")

    def zgemm_grouped(self, problems):
        """One launch for a ragged list of products.  Each problem is a dict with device tensors A, B, C and
        M, N, K, rsA, csA, rsB, csB, ldc (+ optional conjA, conjB, alpha, beta)."""
        n = len(problems)
        if n == 0:
            return
        arr = (ZgemmProblem * n)()
        for q, p in zip(arr, problems):
            q.A, q.B, q.C = p["A"].data_ptr(), p["B"].data_ptr(), p["C"].data_ptr()
            q.M, q.N, q.K = int(p["M"]), int(p["N"]), int(p["K"])
            q.rsA, q.csA, q.rsB, q.csB, q.ldc = int(p["rsA"]), int(p["csA"]), int(p["rsB"]), int(p["csB"]), int(p["ldc"])
            q.conjA, q.conjB = int(bool(p.get("conjA", False))), int(bool(p.get("conjB", False)))
            q.alpha, q.beta = float(p.get("alpha", 1.0)), float(p.get("beta", 0.0))
        self.check(self.lib.dm_zgemm_grouped(self.h, n, arr), "dm_zgemm_grouped")

    def bit_truncate_max_complex(self, t, prec, prec_max_row):
        """In place on a contiguous complex device tensor whose LAST axis is the row (drift/core/beamtransfer.py:641-646)."""
        ncols = int(t.shape[-1])
        nrows = int(t.numel() // max(ncols, 1))
        assert t.is_contiguous()
        self.check(self.lib.dm_bit_truncate_max_complex(self.h, self.ptr(t), nrows, ncols, ncols, float(prec),
                                                        float(prec_max_row)), "dm_bit_truncate_max_complex")

    def zpotrf(self, A, n, ld, stride=0, batch=1):
        info = (c_int * batch)()
        self.check(self.lib.dm_zpotrf_batched(self.h, n, self.ptr(A), ld, stride, batch, info), "dm_zpotrf_batched")
        return np.array(info[:], dtype=np.int64)

    def ztrsm(self, L, B, n, nrhs, ldl, ldb, conjtrans=False, strideL=0, strideB=0, batch=1):
        self.check(self.lib.dm_ztrsm_left_lower_batched(self.h, n, nrhs, self.ptr(L), ldl, strideL, self.ptr(B), ldb,
                                                        strideB, int(conjtrans), batch), "dm_ztrsm")

    def jacobi_rows(self, Z, rows, cols, gc0, gc1, ld, stride=0, batch=1):
        sigma = self.empty((batch, max(rows, 1)), np.float64)
        sw = c_int(0)
        self.check(self.lib.dm_jacobi_rows_batched(self.h, rows, cols, gc0, gc1, self.ptr(Z), ld, stride, batch,
                                                   self.ptr(sigma), ctypes.byref(sw)), "dm_jacobi_rows_batched")
        return sigma, sw.value

    def jacobi_herm(self, C, n, ldc, strideC=0, batch=1):
        W = self.empty((batch, n, n), np.complex128)
        ev = self.empty((batch, max(n, 1)), np.float64)
        sw = c_int(0)
        self.check(self.lib.dm_jacobi_herm_batched(self.h, n, self.ptr(C), ldc, strideC, self.ptr(W), n, n * n, batch,
                                                   self.ptr(ev), ctypes.byref(sw)), "dm_jacobi_herm_batched")
        return ev, W, sw.value


def _iarr(a):
    a = np.ascontiguousarray(a, dtype=np.int32)
    return a, a.ctypes.data_as(ctypes.POINTER(c_int))


def _larr(a):
    a = np.ascontiguousarray(a, dtype=np.int64)
    return a, a.ctypes.data_as(ctypes.POINTER(c_i64))


def _svd_chain(self, beam_m, noisew, polsvcut, skip_svd_inv=False, max_bytes=None, lmin=None):
    """beam_m: device (nblk, F, T, P, L) c128; noisew: device (F, T) f64.
    Returns dict of device tensors + host nmodes (nblk, F) + sweeps[4].
    ``lmin`` (nblk ints, optional): block i is exactly zero in its columns l < lmin[i] (the m-block of m = lmin[i]) —
    the chains then work on the columns l >= lmin only (dm_svd_chain_lmin); the products come back padded.
    The (m, frequency) chains are independent: when the working set of all of them (augmented matrices,
    their row-mixing temporaries, Gram / eigenvector matrices and the eigensolver's workspace) exceeds
    ``max_bytes`` (default: DRIFTMI_SVD_CHUNK_GB, 96) the frequencies go through the library in slices."""
    nblk, F, T, P, L = [int(x) for x in beam_m.shape]
    K = min(L, T)
    _tm = os.environ.get("DRIFTMI_TIMING") == "1"   # (host-side stopwatch of this call on stderr: allocation / library / copies)
    if _tm:
        import sys
        import time
        self.sync(); _t0 = time.perf_counter(); _tc = 0.0
    out = dict(
        beam_svd=self.empty((nblk, F, K, P, L), np.complex128),
        invbeam_svd=None if skip_svd_inv else self.empty((nblk, F, P, L, K), np.complex128),
        beam_ut=self.empty((nblk, F, K, T), np.complex128),
        singularvalues=self.empty((nblk, F, K), np.float64),
    )
    if max_bytes is None:
        max_bytes = float(os.environ.get("DRIFTMI_SVD_CHUNK_GB", "96")) * (1 << 30)
    per_chain = 16.0 * (2.0 * T * (P * L + T) + 16.0 * T * T)
    fc = max(1, min(F, int(max_bytes // max(per_chain * max(nblk, 1), 1.0))))
    nmodes_all = np.zeros((nblk, F), dtype=np.int64)
    sweeps_all = [0, 0, 0, 0]
    lm = None
    if lmin is not None:
        lm_keep, lm = _iarr(np.clip(np.asarray(lmin, dtype=np.int64), 0, L - 1))
        if len(lm_keep) != nblk:
            raise ValueError("svd_chain: one lmin per block expected")
    for f0 in range(0, F, fc):
        f1 = min(F, f0 + fc)
        whole = f0 == 0 and f1 == F
        bm = beam_m if whole else beam_m[:, f0:f1].contiguous()
        nw = noisew if whole else noisew[f0:f1].contiguous()
        o = out if whole else dict(
            beam_svd=self.empty((nblk, f1 - f0, K, P, L), np.complex128),
            invbeam_svd=None if skip_svd_inv else self.empty((nblk, f1 - f0, P, L, K), np.complex128),
            beam_ut=self.empty((nblk, f1 - f0, K, T), np.complex128),
            singularvalues=self.empty((nblk, f1 - f0, K), np.float64))
        nmodes = (c_int * max(nblk * (f1 - f0), 1))()
        sweeps = (c_int * 4)()
        if _tm:
            self.sync(); _t1 = time.perf_counter()
        rc = self.lib.dm_svd_chain_lmin(self.h, nblk, f1 - f0, T, P, L, lm, self.ptr(bm), self.ptr(nw), float(polsvcut),
                                        self.ptr(o["beam_svd"]), self.ptr(o["invbeam_svd"]), self.ptr(o["beam_ut"]),
                                        self.ptr(o["singularvalues"]), nmodes, sweeps)
        if _tm:
            self.sync(); _tc += time.perf_counter() - _t1
        self.check(rc, "dm_svd_chain_lmin")
        nmodes_all[:, f0:f1] = np.array(nmodes[: nblk * (f1 - f0)], dtype=np.int64).reshape(nblk, f1 - f0)
        sweeps_all = [max(a, int(b)) for a, b in zip(sweeps_all, sweeps)]
        if not whole:
            for k in ("beam_svd", "invbeam_svd", "beam_ut", "singularvalues"):
                if out[k] is not None:
                    out[k][:, f0:f1].copy_(o[k])
            del o, bm
    out["nmodes"] = nmodes_all
    out["sweeps"] = sweeps_all
    if _tm:
        self.sync()
        _tt = time.perf_counter() - _t0
        sys.stderr.write("[timing] svd_chain %d blocks, %d slice(s): %.3f s, of it %.3f s in the library, %.3f s allocation + copies\n"
                         % (nblk, -(-F // fc), _tt, _tc, _tt - _tc))
    return out


def _block_offsets(ndofs):
    ndofs = np.asarray(ndofs, dtype=np.int64)
    off = np.concatenate([[0], np.cumsum(ndofs * ndofs)])
    return off[:-1].copy(), int(off[-1])


def _project_cov(self, beam_svd, svnum, cl_pfl, out, out_off, npol=None, polmask=None, l0=None, zero_first=True,
                 symmetric=False):
    """``symmetric``: the caller has checked cl[p,q,f,f',l] == cl[p,q,f',f,l]; only then may the library
    form the frequency blocks f' >= f alone and mirror them (the reference accepts any array)."""
    nblk, F, K, P, L = [int(x) for x in beam_svd.shape]
    sv, svp = _iarr(svnum)
    off, offp = _larr(out_off)
    npol = P if npol is None else int(npol)
    pm = pmp = None
    if polmask is not None:
        pm, pmp = _iarr(polmask)
    l0a = l0p = None
    if l0 is not None:
        l0a, l0p = _iarr(l0)
    rc = self.lib.dm_project_cov(self.h, nblk, F, K, P, L, self.ptr(beam_svd), svp, l0p, self.ptr(cl_pfl), npol, pmp,
                                 self.ptr(out), offp, int(bool(zero_first)) | (2 if symmetric else 0))
    self.check(rc, "dm_project_cov")


def _fisher(self, beam_svd, svnum, l0, cl_bands, evecs, evecs_off, nmodes, evals, evals_off, cl_symmetric=False):
    """Per-m Fisher matrices (nblk, nbands, nbands) complex; see dm_fisher in include/driftmi.h."""
    nblk, F, K, P, L = [int(x) for x in beam_svd.shape]
    nbands = int(cl_bands.shape[0])
    sv, svp = _iarr(svnum)
    l0a, l0p = _iarr(l0)
    eo, eop = _larr(evecs_off)
    vo, vop = _larr(evals_off)
    nm, nmp = _iarr(nmodes)
    out = self.empty((nblk, nbands, nbands), np.complex128)
    rc = self.lib.dm_fisher(self.h, nblk, F, K, P, L, self.ptr(beam_svd), svp, l0p, nbands, self.ptr(cl_bands),
                            self.ptr(evecs), eop, nmp, self.ptr(evals), vop, self.ptr(out), int(bool(cl_symmetric)))
    self.check(rc, "dm_fisher")
    return out


Context.fisher = _fisher


def _project_diag(self, beam_ut, svnum, dmat, out, out_off, alpha=1.0, accumulate=False):
    nblk, F, K, T = [int(x) for x in beam_ut.shape]
    sv, svp = _iarr(svnum)
    off, offp = _larr(out_off)
    rc = self.lib.dm_project_diag(self.h, nblk, F, K, T, self.ptr(beam_ut), svp, self.ptr(dmat), float(alpha),
                                  self.ptr(out), offp, int(accumulate))
    self.check(rc, "dm_project_diag")


def _regularise(self, mats, ndofs, off, reg):
    n, np_ = _iarr(ndofs)
    o, op = _larr(off)
    self.check(self.lib.dm_regularise(self.h, len(n), np_, self.ptr(mats), op, float(reg)), "dm_regularise")


def _eigh_gen(self, A, B, ndofs, off, cut=None):
    """A, B: flat device c128 buffers holding the (n_b x n_b) blocks at `off`.  Destroys both.
    cut: None, ("upper", thr) — only the modes with eigenvalue >= thr (rows i >= searchsorted(evals, thr))
    are formed — or ("lower", thr) for the rows below; the other rows of a block are zero.
    Returns (evals flat device f64 [offsets evoff], evoff, evecs flat device c128 [same off], add_const, sweeps);
    the number of rows formed per block is left in ``self.last_nkeep``."""
    n, np_ = _iarr(ndofs)
    o, op = _larr(off)
    evoff = np.concatenate([[0], np.cumsum(n.astype(np.int64))])
    eo, eop = _larr(evoff[:-1])
    evals = self.empty((max(int(evoff[-1]), 1),), np.float64)
    evecs = self.empty((max(int(A.numel()), 1),), np.complex128)
    ac = (c_dbl * max(len(n), 1))()
    nk = (c_int * max(len(n), 1))()
    sw = c_int(0)
    mode, thr = 0, 0.0
    if cut is not None:
        mode = {"upper": 1, "lower": 2}[cut[0]]
        thr = float(cut[1])
    rc = self.lib.dm_eigh_gen(self.h, len(n), np_, self.ptr(A), self.ptr(B), op, self.ptr(evals), eop,
                              self.ptr(evecs), ac, ctypes.byref(sw), mode, thr, nk)
    self.check(rc, "dm_eigh_gen")
    self.last_nkeep = np.array(nk[: len(n)], dtype=np.int64)
    return evals, evoff, evecs, np.array(ac[: len(n)], dtype=np.float64), sw.value


def _kl_m(self, beam_svd, beam_ut, svnum, l0, cl_sg, sg_mask, sg_sym, cl_fg, fg_mask, fg_sym, npower, noise_scale, regulariser,
          cut=None):
    """dm_kl_m: covariances + generalised eigenproblem of a batch of m-blocks in one library call.  Returns
    (evals flat, evoff, evecs flat, off, add_const, nkeep)."""
    nblk, F, K, P, L = [int(x) for x in beam_svd.shape]
    T = int(beam_ut.shape[-1])
    sv, svp = _iarr(svnum)
    ndofs = np.asarray(svnum).reshape(nblk, F).sum(axis=1).astype(np.int64)
    off, tot = _block_offsets(ndofs)
    evoff = np.concatenate([[0], np.cumsum(ndofs)])
    o, op = _larr(off)
    eo, eop = _larr(evoff[:-1])
    l0a, l0p = (None, None) if l0 is None else _iarr(l0)
    sm, smp = (None, None) if sg_mask is None else _iarr(sg_mask)
    fm, fmp = (None, None) if fg_mask is None else _iarr(fg_mask)
    evals = self.empty((max(int(evoff[-1]), 1),), np.float64)
    evecs = self.empty((max(tot, 1),), np.complex128)
    ac = (c_dbl * max(nblk, 1))()
    nk = (c_int * max(nblk, 1))()
    mode, thr = (0, 0.0) if cut is None else ({"upper": 1, "lower": 2}[cut[0]], float(cut[1]))
    rc = self.lib.dm_kl_m(self.h, nblk, F, K, P, L, T, self.ptr(beam_svd), self.ptr(beam_ut), svp, l0p, self.ptr(cl_sg), smp,
                          int(bool(sg_sym)), self.ptr(cl_fg), fmp, int(bool(fg_sym)), self.ptr(npower), float(noise_scale),
                          float(regulariser), mode, thr, self.ptr(evals), eop, self.ptr(evecs), op, ac, nk)
    self.check(rc, "dm_kl_m")
    return evals, evoff, evecs, off, np.array(ac[:nblk]), np.array(nk[:nblk], dtype=np.int64)


def _doublekl_m(self, beam_svd, beam_ut, svnum, l0, cl_sg, sg_mask, sg_sym, cl_fg, fg_mask, fg_sym, npower, floor_scale,
                regulariser, foreground_threshold, cut=None):
    """dm_doublekl_m.  Returns (f_evals flat, evals flat, evoff, modes flat, off, nmodes, nkeep, add_const)."""
    nblk, F, K, P, L = [int(x) for x in beam_svd.shape]
    T = int(beam_ut.shape[-1])
    sv, svp = _iarr(svnum)
    ndofs = np.asarray(svnum).reshape(nblk, F).sum(axis=1).astype(np.int64)
    off, tot = _block_offsets(ndofs)
    evoff = np.concatenate([[0], np.cumsum(ndofs)])
    o, op = _larr(off)
    eo, eop = _larr(evoff[:-1])
    l0a, l0p = (None, None) if l0 is None else _iarr(l0)
    sm, smp = (None, None) if sg_mask is None else _iarr(sg_mask)
    fm, fmp = (None, None) if fg_mask is None else _iarr(fg_mask)
    f_evals = self.empty((max(int(evoff[-1]), 1),), np.float64)
    evals = self.zeros((max(int(evoff[-1]), 1),), np.float64)
    modes = self.zeros((max(tot, 1),), np.complex128)
    ac = (c_dbl * max(nblk, 1))()
    nm = (c_int * max(nblk, 1))()
    nk = (c_int * max(nblk, 1))()
    mode, thr = (0, 0.0) if cut is None else ({"upper": 1, "lower": 2}[cut[0]], float(cut[1]))
    rc = self.lib.dm_doublekl_m(self.h, nblk, F, K, P, L, T, self.ptr(beam_svd), self.ptr(beam_ut), svp, l0p, self.ptr(cl_sg), smp,
                                int(bool(sg_sym)), self.ptr(cl_fg), fmp, int(bool(fg_sym)), self.ptr(npower), float(floor_scale),
                                float(regulariser), float(foreground_threshold), mode, thr, self.ptr(f_evals), self.ptr(evals),
                                eop, self.ptr(modes), op, nm, nk, ac)
    self.check(rc, "dm_doublekl_m")
    return (f_evals, evals, evoff, modes, off, np.array(nm[:nblk], dtype=np.int64), np.array(nk[:nblk], dtype=np.int64),
            np.array(ac[:nblk]))


Context.kl_m = _kl_m
Context.doublekl_m = _doublekl_m
Context.svd_chain = _svd_chain
Context.project_cov = _project_cov
Context.project_diag = _project_diag
Context.regularise = _regularise
Context.eigh_gen = _eigh_gen
block_offsets = _block_offsets


def _darr(a):
    a = np.ascontiguousarray(a, dtype=np.float64)
    return a, a.ctypes.data_as(ctypes.POINTER(c_dbl))


def _bt_beam_cyl(self, nside, cth, sth, frame, kind, tab, fwhm_ns, out):
    c, cp = _darr(cth)
    s_, sp = _darr(sth)
    fr, frp = _darr(frame)
    tx, txp = _darr(tab[0])
    ty, typ = _darr(tab[1])
    t2, t2p = _darr(tab[2])
    rc = self.lib.dm_bt_beam_cyl(self.h, int(nside), cp, sp, frp, int(kind), txp, typ, t2p, len(tx), float(fwhm_ns),
                                 self.ptr(out))
    self.check(rc, "dm_bt_beam_cyl")


def _bt_beams_cyl(self, nside, cth, sth, frame, specs, out, rows):
    """Evaluate the cylinder patterns `specs` = [(kind, (x, y, y2), fwhm_ns), ...] into the rows `rows` of the 2-D
    float64 device tensor `out` — consecutive rows go up in one dm_bt_beams_cyl call each."""
    if not specs:
        return
    c, cp = _darr(cth)
    s_, sp = _darr(sth)
    fr, frp = _darr(frame)
    stride = int(out.stride(0))
    i = 0
    while i < len(specs):
        j = i + 1
        while j < len(specs) and rows[j] == rows[j - 1] + 1:
            j += 1
        kinds, kp = _iarr([sp_[0] for sp_ in specs[i:j]])
        fw, fwp = _darr([sp_[2] for sp_ in specs[i:j]])
        offs = np.concatenate([[0], np.cumsum([len(sp_[1][0]) for sp_ in specs[i:j]])])
        of, ofp = _iarr(offs)
        tx, txp = _darr(np.concatenate([sp_[1][0] for sp_ in specs[i:j]]))
        ty, typ = _darr(np.concatenate([sp_[1][1] for sp_ in specs[i:j]]))
        t2, t2p = _darr(np.concatenate([sp_[1][2] for sp_ in specs[i:j]]))
        rc = self.lib.dm_bt_beams_cyl(self.h, int(nside), cp, sp, frp, j - i, kp, txp, typ, t2p, ofp, fwp,
                                      self.ptr(out[rows[i]]), stride)
        self.check(rc, "dm_bt_beams_cyl")
        i = j


def _bt_maps(self, nside, cth, sth, frame, polarised, beams, uv, bi, bj, maps):
    c, cp = _darr(cth)
    s_, sp = _darr(sth)
    fr, frp = _darr(frame)
    u, up = _darr(uv)
    i_, ip = _iarr(bi)
    j_, jp = _iarr(bj)
    # complex128 beams take the complex-pattern kernels (second beam conjugated, |b|^2 solid angles)
    fn = self.lib.dm_bt_maps_c if beams.is_complex() else self.lib.dm_bt_maps
    rc = fn(self.h, int(nside), cp, sp, frp, int(bool(polarised)), int(beams.shape[0]),
            self.ptr(beams), len(i_), up, ip, jp, self.ptr(maps))
    self.check(rc, "dm_bt_maps")


def _bt_sht(self, nside, cth, sth, polarised, lside, mmax, lmax_grp, F, B, col_f, col_b, col_lmax, maps, beam_m,
            m_range=None, niter=0, ring_w=None):
    """m_range = (m_lo, m_hi): only those m-blocks are produced and `beam_m` has m_hi - m_lo + 1 of them.
    niter / ring_w: healpy.map2alm's `iter` and ring weights (see dm_bt_sht_opts)."""
    c, cp = _darr(cth)
    s_, sp = _darr(sth)
    f_, fp = _iarr(col_f)
    b_, bp = _iarr(col_b)
    l_, lp = _iarr(col_lmax)
    if niter or ring_w is not None:
        m_lo, m_hi = (0, int(mmax)) if m_range is None else (int(m_range[0]), int(m_range[1]))
        w, wp = (None, None) if ring_w is None else _darr(ring_w)
        if w is not None and w.size != 4 * int(nside) - 1:
            raise ValueError("ring weights: need one per ring (4 nside - 1 = %d), got %d" % (4 * int(nside) - 1, w.size))
        rc = self.lib.dm_bt_sht_opts(self.h, int(nside), cp, sp, int(bool(polarised)), int(lside), m_lo, m_hi,
                                     int(lmax_grp), int(F), int(B), len(f_), fp, bp, lp, self.ptr(maps), self.ptr(beam_m),
                                     int(niter), wp)
        self.check(rc, "dm_bt_sht_opts")
        return
    if m_range is None:
        rc = self.lib.dm_bt_sht(self.h, int(nside), cp, sp, int(bool(polarised)), int(lside), int(mmax), int(lmax_grp),
                                int(F), int(B), len(f_), fp, bp, lp, self.ptr(maps), self.ptr(beam_m))
    else:
        rc = self.lib.dm_bt_sht_range(self.h, int(nside), cp, sp, int(bool(polarised)), int(lside), int(m_range[0]),
                                      int(m_range[1]), int(lmax_grp), int(F), int(B), len(f_), fp, bp, lp,
                                      self.ptr(maps), self.ptr(beam_m))
    self.check(rc, "dm_bt_sht")


def _bt_columns(self, nside, cth, sth, frame, polarised, beams, uv, bi, bj, lside, mmax, lmax_grp, F, B, col_f, col_b,
                col_lmax, beam_m, m_range=None, ring_w=None, niter=0):
    """Beams -> beam_m rows of the given columns without materialising the Stokes maps (dm_bt_columns); with
    niter > 0 healpy's `iter` refinements in harmonic space (dm_bt_columns_iter)."""
    c, cp = _darr(cth)
    s_, sp = _darr(sth)
    fr, frp = _darr(frame)
    u, up = _darr(uv)
    i_, ip = _iarr(bi)
    j_, jp = _iarr(bj)
    f_, fp = _iarr(col_f)
    b_, bp = _iarr(col_b)
    l_, lp = _iarr(col_lmax)
    m_lo, m_hi = (0, int(mmax)) if m_range is None else (int(m_range[0]), int(m_range[1]))
    w, wp = (None, None) if ring_w is None else _darr(ring_w)
    # complex field patterns (a complex128 beams tensor) take the entry that forms _construct_pol_complex in the kernels
    if niter:
        rc = self.lib.dm_bt_columns_iter(self.h, int(nside), cp, sp, frp, int(bool(polarised)), int(beams.shape[0]),
                                         self.ptr(beams), int(beams.is_complex()), len(i_), up, ip, jp, int(lside), m_lo, m_hi,
                                         int(lmax_grp), int(F), int(B), fp, bp, lp, self.ptr(beam_m), wp, int(niter))
        self.check(rc, "dm_bt_columns_iter")
        return
    fn = self.lib.dm_bt_columns_c if beams.is_complex() else self.lib.dm_bt_columns
    rc = fn(self.h, int(nside), cp, sp, frp, int(bool(polarised)), int(beams.shape[0]), self.ptr(beams),
            len(i_), up, ip, jp, int(lside), m_lo, m_hi, int(lmax_grp), int(F), int(B), fp, bp, lp,
            self.ptr(beam_m), wp)
    self.check(rc, "dm_bt_columns")


def bt_alias_info(nside, cth, sth, polarised, lmax_grp):
    """(alias rings per cap, mcut) of the harmonic-space refinement for one nside group (dm_bt_alias_info; host only)."""
    c, cp = _darr(cth)
    s_, sp = _darr(sth)
    nr, mc = c_int(0), c_int(0)
    rc = load().dm_bt_alias_info(int(nside), cp, sp, int(bool(polarised)), int(lmax_grp), ctypes.byref(nr), ctypes.byref(mc))
    if rc != 0:
        raise DriftMIError("dm_bt_alias_info failed (%d)" % rc)
    return nr.value, mc.value


Context.bt_columns = _bt_columns
Context.bt_beam_cyl = _bt_beam_cyl
Context.bt_beams_cyl = _bt_beams_cyl
Context.bt_maps = _bt_maps
Context.bt_sht = _bt_sht


PROF_CLASSES = ["zgemm_grouped", "gemm_grouped_realB", "jac_gram", "jac_inner", "jac_apply", "dgemm_grouped",
                "trd_symv", "trd_wx",  # these two report algorithmic BYTES in the "flops" field (HBM-bound kernels)
                "sb_panel_qr", "sb_chase", "sb_q2_apply",   # two-stage tridiagonalisation (fp64 VALU)
                "zgemm_cov",                                  # gathered-B ZGEMM: the covariance projections
                # extended classes (profiling level 2, time only)
                "bt_ring", "bt_other", "trd_small", "dc", "chol_solve", "util", "eig_other", "svd_other",
                "unused20", "unused21", "unused22", "unused23"]
PROF_NCLASS = len(PROF_CLASSES)


def _prof_reset(self, enable=True):
    """enable: False / True (the MFMA and HBM-bound classes) / 2 (every kernel of the path: the extended classes too)."""
    self.check(self.lib.dm_prof_reset(self.h, int(enable)), "dm_prof_reset")
    self._prof_level = int(enable)


def _prof_enabled(self):
    return bool(getattr(self, "_prof_level", 0))


def _prof_report(self):
    ms = (c_dbl * PROF_NCLASS)()
    fl = (c_dbl * PROF_NCLASS)()
    ln = (ctypes.c_longlong * PROF_NCLASS)()
    self.check(self.lib.dm_prof_report(self.h, ms, fl, ln), "dm_prof_report")
    return {name: dict(ms=ms[i], flops=fl[i], launches=int(ln[i])) for i, name in enumerate(PROF_CLASSES) if ln[i] > 0}


Context.prof_reset = _prof_reset
Context.prof_enabled = _prof_enabled
Context.prof_report = _prof_report


def _herm_eig(self, C, n, ldc, strideC=0, batch=1):
    W = self.empty((batch, n, n), np.complex128)
    ev = self.empty((batch, max(n, 1)), np.float64)
    rc = self.lib.dm_herm_eig_batched(self.h, n, self.ptr(C), ldc, strideC, self.ptr(W), n, n * n, batch, self.ptr(ev))
    self.check(rc, "dm_herm_eig_batched")
    return ev, W


Context.herm_eig = _herm_eig
