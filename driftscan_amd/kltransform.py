"""Karhunen-Loeve transform per m, GPU-backed.

Mirrors ``drift.core.kltransform.KLTransform`` (drift/core/kltransform.py:146-911):
same constructor, config properties, file names and datasets, ``sn_covariance``,
``transform_save``, ``generate``, ``evals_all``, ``modes_m`` & co.  The covariance
projections and the generalised eigenproblem run on the GPU for all of this rank's
m-blocks at once (``dm_project_cov`` / ``dm_project_diag`` / ``dm_regularise`` /
``dm_eigh_gen``).
"""
import logging
import os
import time

import numpy as np

from . import config, parallel, skymodel, storage, util
from ._lib import block_offsets
from .device import get_context

logger = logging.getLogger(__name__)


def eigh_gen(A, B, message=""):
    """Generalised Hermitian-definite eigenproblem A v = lambda B v on the GPU with the
    reference's all-zero shortcut and non-positive-definite rescue (kltransform.py:55-121).
    Returns (evals ascending, evecs as columns, add_const)."""
    ctx = get_context()
    A = np.ascontiguousarray(A, dtype=np.complex128)
    B = np.ascontiguousarray(B, dtype=np.complex128)
    n = A.shape[0]
    off, tot = block_offsets([n])
    evals, evoff, evecs, ac, _ = ctx.eigh_gen(ctx.to_device(A.ravel()), ctx.to_device(B.ravel()), [n], off)
    E = evecs[: n * n].cpu().numpy().reshape(n, n)
    return evals[:n].cpu().numpy(), E.T.conj(), float(ac[0])


class KLTransform(config.Reader):
    subset = config.Property(proptype=config.truthy, default=True, key="subset")
    inverse = config.Property(proptype=config.truthy, default=False, key="inverse")
    threshold = config.Property(proptype=float, default=0.1, key="threshold")
    _foreground_regulariser = config.Property(proptype=float, default=1e-14, key="regulariser")
    use_thermal = config.Property(proptype=config.truthy, default=True)
    use_foregrounds = config.Property(proptype=config.truthy, default=True)
    use_polarised = config.Property(proptype=config.truthy, default=True)
    pol_length = config.Property(proptype=float, default=None)
    # MI355X-side knob: device memory budget for one batch of m-blocks
    kl_chunk_gb = config.Property(proptype=float, default=48.0)

    evdir = ""
    _cut_ok = True  # subclasses whose _transform needs every mode (DoubleKL) turn the early cut off
    _cvfg = None
    _cvsg = None

    @property
    def _evfile(self):
        return self.evdir + "/ev_m_" + util.natpattern(self.telescope.mmax) + ".hdf5"

    def __init__(self, bt, subdir=None):
        self.beamtransfer = bt
        self.telescope = bt.telescope
        subdir = "ev" if subdir is None else subdir
        self.evdir = bt.directory + "/" + subdir
        if parallel.io_root() and not os.path.exists(self.evdir):
            os.makedirs(self.evdir)
        parallel.barrier()

    # ---- sky covariances (kltransform.py:203-256) ------------------------------------
    def foreground(self):
        if self._cvfg is None:
            npol = self.telescope.num_pol_sky
            if npol not in (1, 3, 4):
                raise Exception("Can only handle unpolarised only (num_pol_sky = 1), or I, Q and U (num_pol_sky = 3).")
            if self.use_polarised:
                self._cvfg = skymodel.foreground_model(self.telescope.lmax, self.telescope.frequencies, npol,
                                                       pol_length=self.pol_length)
            else:
                self._cvfg = skymodel.foreground_model(self.telescope.lmax, self.telescope.frequencies, npol,
                                                       pol_frac=0.0)
        return self._cvfg

    def signal(self):
        if self._cvsg is None:
            npol = self.telescope.num_pol_sky
            if npol not in (1, 3, 4):
                raise Exception("Can only handle unpolarised only (num_pol_sky = 1), or I, Q and U (num_pol_sky = 3).")
            self._cvsg = skymodel.im21cm_model(self.telescope.lmax, self.telescope.frequencies, npol)
        return self._cvsg

    # ---- covariances in the SVD basis ---------------------------------------------------
    def _npower(self, nc):
        tel = self.telescope
        bl = np.arange(tel.npairs)
        bl = np.concatenate((bl, bl))
        return nc * np.asarray(tel.noisepower(bl[np.newaxis, :], np.arange(tel.nfreq)[:, np.newaxis])).reshape(
            tel.nfreq, self.beamtransfer.ntel)

    def _npower_device(self, nc):
        """_npower(nc) on the device, uploaded once per (context, nc) — see BeamTransfer._noisew_device."""
        ctx = get_context()
        key = (id(ctx), float(nc))
        cache = self.__dict__.setdefault("_npower_dev", {})
        if key not in cache:
            if len(cache) > 8:
                cache.clear()
            cache[key] = ctx.to_device(self._npower(nc))
        return cache[key]

    def sn_covariance_device(self, ms):
        """Signal and noise covariances of several m at once, left on the device.
        Returns (S, N, ndofs, off): flat complex buffers with block b at off[b]."""
        if not (self.use_foregrounds or self.use_thermal):
            raise Exception("Either `use_thermal` or `use_foregrounds`, or both must be True.")
        ctx = get_context()
        bt = self.beamtransfer
        ndofs = np.array([int(bt.ndof(mi)) for mi in ms], dtype=np.int64)
        off, tot = block_offsets(ndofs)
        S = ctx.empty((max(tot, 1),), np.complex128)
        N = ctx.empty((max(tot, 1),), np.complex128)
        bt.project_matrix_sky_to_svd_device(ms, self.signal(), S, off)
        if self.use_foregrounds:
            bt.project_matrix_sky_to_svd_device(ms, self.foreground(), N, off)
        else:
            N.zero_()
        # small diagonal to regularise the noise matrix (kltransform.py:288-290)
        ctx.regularise(N, ndofs, off, self._foreground_regulariser)
        # even without thermal noise keep a 1 mK floor (kltransform.py:292-296)
        nc = 1.0 if self.use_thermal else (1e-3 / self.telescope.tsys_flat) ** 2
        but = bt._stacked_products(ms, "beam_ut")
        svnum = np.stack([bt._svd_num(mi)[0] for mi in ms])
        ctx.project_diag(but, svnum, self._npower_device(nc), N, off, alpha=1.0, accumulate=True)
        return S, N, ndofs, off

    def sn_covariance(self, mi):
        """(S, N) of one m as numpy arrays (kltransform.py:258-308)."""
        S, N, ndofs, off = self.sn_covariance_device([mi])
        get_context().sync()
        n = int(ndofs[0])
        return S[: n * n].cpu().numpy().reshape(n, n), N[: n * n].cpu().numpy().reshape(n, n)

    # ---- the transform -------------------------------------------------------------------
    def _transform_batch(self, ms, to_host=True):
        """KL modes of several m: list of (evals, evecs[rows = modes], inv, evextra).
        With ``to_host=False`` the eigenvectors stay on the device (views into one flat tensor)."""
        ctx = get_context()
        S, N, ndofs, off = self.sn_covariance_device(ms)
        # With `subset` only the modes above the S/N threshold are ever saved (kltransform.py:388-398)
        # and without `inverse` nothing else needs the discarded ones: they are not back-transformed
        # (their rows of the returned matrix are zero; `evals` is complete either way).
        cut = ("upper", self.threshold) if (self.subset and not self.inverse and self._cut_ok) else None
        need = 8.5 * 16.0 * float((np.asarray(ndofs, dtype=np.float64) ** 2).sum())  # eigensolver workspace, one arena
        if need > (64 << 30):
            # a CHIME-sized block (ndof 32 576: ~140 GB): nothing else may stay on the card — the SVD products of
            # the batch are re-read from their files when they are needed again, the C_l tables re-uploaded
            import torch

            from .beamtransfer import BeamTransfer

            for mi in ms:
                self.beamtransfer._dev.pop(mi, None)
            self.beamtransfer.__dict__.pop("_stack_memo", None)
            BeamTransfer._clcache.clear()
            ctx.sync()
            torch.cuda.empty_cache()
            ctx.workspace_reset(int(need))
        Nk = N.clone() if self.inverse else None  # eigh_gen destroys its inputs; the inverse is N E^H (below)
        evals, evoff, evecs, ac, sweeps = ctx.eigh_gen(S, N, ndofs, off, cut=cut)
        ev_h = evals.cpu().numpy()
        inv = None
        if self.inverse:
            inv = self._inverse_device(ctx, evecs, Nk, ndofs, off, ac, [ev_h[evoff[i] : evoff[i + 1]] for i in range(len(ms))])
        if not to_host:
            nn = [int(n) for n in ndofs]
            sq = [n * n for n in nn]
            if all(int(off[i + 1]) == int(off[i]) + sq[i] for i in range(len(ms) - 1)):
                # blocks stored back to back (the usual case): the per-m views come out of one split call each — a
                # few hundred tensor slicings in Python are a millisecond at the end of every batch
                import torch

                o0 = int(off[0]) if len(ms) else 0
                evs = torch.split(evals[int(evoff[0]) : int(evoff[0]) + sum(nn)], nn)
                vecs = torch.split(evecs[o0 : o0 + sum(sq)], sq)
                invs = None if inv is None else torch.split(inv[o0 : o0 + sum(sq)], sq)
                return [(evs[i], vecs[i].view(nn[i], nn[i]), None if invs is None else invs[i].view(nn[i], nn[i]),
                         {"ac": float(ac[i])}) for i in range(len(ms))]
            return [(evals[evoff[i] : evoff[i] + nn[i]],
                     evecs[off[i] : off[i] + sq[i]].view(nn[i], nn[i]),
                     None if inv is None else inv[off[i] : off[i] + sq[i]].view(nn[i], nn[i]),
                     {"ac": float(ac[i])}) for i in range(len(ms))]
        out = []
        for i, mi in enumerate(ms):
            n = int(ndofs[i])
            if n == 0:
                out.append((np.array([]), np.array([[]]), np.array([[]]), {"ac": 0.0}))
                continue
            E = ctx.to_host(evecs[off[i] : off[i] + n * n]).reshape(n, n)
            inv_i = None if inv is None else ctx.to_host(inv[off[i] : off[i] + n * n]).reshape(n, n)
            out.append((ev_h[evoff[i] : evoff[i] + n].copy(), E, inv_i, {"ac": float(ac[i])}))
        return out

    @staticmethod
    def _inverse_device(ctx, evecs, Nk, ndofs, off, ac, evals_host):
        """`inv_gen(evecs).T` of the reference (kltransform.py:124-143, :346-347) for every block, on the device:
        the modes satisfy E (N + ac I) E^H = I, so E^-1 = (N + ac I) E^H and its transpose is conj(E (N + ac I)) —
        one grouped product instead of an LU inversion per m on the host.  The all-zero shortcut of eigh_gen
        (kltransform.py:81-85) returns the identity, whose inverse is the identity."""
        import torch

        inv = ctx.empty((max(int(evecs.numel()), 1),), np.complex128)
        probs = []
        for i, n in enumerate(int(x) for x in ndofs):
            if n == 0:
                continue
            blk = slice(int(off[i]), int(off[i]) + n * n)
            if not np.any(evals_host[i]):
                inv[blk].view(n, n).copy_(torch.eye(n, dtype=torch.complex128, device=inv.device))
                continue
            if ac[i] != 0.0:
                Nk[blk].view(n, n).diagonal().add_(float(ac[i]))
            probs.append(dict(A=evecs[blk], B=Nk[blk], C=inv[blk], M=n, N=n, K=n, rsA=n, csA=1, rsB=n, csB=1, ldc=n,
                              conjA=True, conjB=True))
        ctx.zgemm_grouped(probs)
        return inv

    def _transform_m(self, mi):
        return self._transform_batch([mi])[0]

    def _save(self, mi, evals, evecs, inv, evextra, nside=None):
        """Write ev_m_<m>.hdf5 (kltransform.py:377-421).  `nside` (= ndof of the m) is looked up by the
        submitting thread: the accessors behind it keep a last-call cache that the writer threads must not share."""
        if nside is None:
            nside = int(self.beamtransfer.ndof(mi))
        with storage.File(self._evfile % mi, "w") as f:
            f.attrs["m"] = mi
            f.attrs["SUBSET"] = bool(self.subset)
            evalsf = np.zeros(nside, dtype=np.float64)
            if evals.size != 0:
                evalsf[-evals.size :] = evals
            f.create_dataset("evals_full", data=evalsf)
            if self.subset:
                i_ev = np.searchsorted(evals, self.threshold)
                evals = evals[i_ev:]
                evecs = evecs[i_ev:]
                logger.info("Modes with S/N > %f: %i of %i" % (self.threshold, evals.size, evalsf.size))
            f.create_dataset("evals", data=evals)
            f.create_dataset("evecs", data=evecs)
            f.attrs["num_modes"] = evals.size
            if self.inverse:
                if self.subset:
                    inv = inv[i_ev:]
                f.create_dataset("evinv", data=inv)
            self._ev_save_hook(f, evextra)
        return evals, evecs

    def transform_save(self, mi):
        evals, evecs, inv, evextra = self._transform_m(mi)
        return self._save(mi, evals, evecs, inv, evextra)

    def _ev_save_hook(self, f, evextra):
        ac = evextra["ac"]
        if ac != 0.0:
            f.attrs["add_const"] = ac
            f.attrs["FLAGS"] = "NotPositiveDefinite"
        else:
            f.attrs["FLAGS"] = "Normal"

    def _batches(self, ms):
        """Split this rank's m list so that the S, N, L, C, W, E buffers of a batch fit the budget."""
        budget = self.kl_chunk_gb * (1 << 30)
        out, cur, used = [], [], 0.0
        for mi in ms:
            n = float(self.beamtransfer.ndof(mi))
            need = 16.0 * n * n * 16.0  # S, N, E, L, C, W, V, Z, recorded rotations, staging
            if cur and used + need > budget:
                out.append(cur)
                cur, used = [], 0.0
            cur.append(mi)
            used += need
        if cur:
            out.append(cur)
        if len(out) > 1:
            # the same number of batches with about equal counts (still within the budget: blocks of neighbouring m have
            # about the same ndof) instead of full batches and a short one — see BeamTransfer._svd_batch_lists
            ms = [mi for b in out for mi in b]
            base, rem = divmod(len(ms), len(out))
            even, c0 = [], 0
            for k in range(len(out)):
                n = base + (1 if k < rem else 0)
                even.append(ms[c0 : c0 + n])
                c0 += n
            if all(sum(16.0 * float(self.beamtransfer.ndof(mi)) ** 2 * 16.0 for mi in b) <= budget for b in even):
                out = even
        return out

    def generate_ms(self, ms, regen=False):
        """KL-transform the given m (of this rank) and hand the files to the writer pool: the per-batch step that
        `ProductManager.generate` runs while the SVD products of these m are still resident.  No barrier, no collect."""
        done = self.__dict__.setdefault("_done", set())
        full = self.__dict__.setdefault("_evals_full_mem", {})
        todo = [mi for mi in ms if mi not in done and (regen or not os.path.exists(self._evfile % mi))]
        cache = self.__dict__.get("_mode_cache")   # set by whoever wants the modes of the batch right away (PS estimation)
        ctx = get_context()
        for batch in self._batches(todo):
            # The modes stay on the device: whoever wants them right away (the Fisher estimators, through the mode cache)
            # reads them there, and the files get their host copies from the writer pool's copy thread (storage.Deferred)
            # while the next batch is computed — round 3 copied every eigenvector matrix to the host here and the estimator
            # sent it back (2 x 15 GB per configs[3] share through the driver thread).
            results = self._transform_batch(batch, to_host=False)
            event = None if storage.discard() else ctx.record_event()

            def on_host(x):
                if x is None or isinstance(x, np.ndarray):
                    return x
                return ctx.defer_host(x, event) if x.numel() else x.cpu().numpy()

            for mi, res in zip(batch, results):
                nside = int(self.beamtransfer.ndof(mi))
                evf = np.zeros(nside)
                ev = res[0] if isinstance(res[0], np.ndarray) else res[0].cpu().numpy()
                if ev.size:
                    evf[-ev.size :] = ev
                full[mi] = evf
                self.__dict__.setdefault("_extra_mem", {})[mi] = res[3]
                done.add(mi)
                if cache is not None:
                    i_ev = int(np.searchsorted(ev, self.threshold)) if self.subset else 0
                    cache[mi] = (ev[i_ev:], res[1][i_ev:]) if ev.size else (ev, res[1])
                # written in the background while the next batch is computed
                if not storage.discard():
                    storage.submit(self._save, mi, ev, on_host(res[1]), on_host(res[2]), res[3], nside)

    def generate(self, regen=False):
        """KL-transform every m of this rank and save (kltransform.py:480-513); m already done by `generate_ms` during
        beam-transfer generation are skipped."""
        st = time.time()
        storage.flush()          # the files of the batches `generate_ms` has handed to the writer pool
        done = self.__dict__.get("_done")
        if regen:
            self.__dict__.pop("_done", None)
        elif done and not storage.discard():
            # like the reference, the FILES decide what is done (kltransform.py:497-501): an m this object transformed
            # earlier whose file has gone since is made again
            for mi in [mi for mi in done if not os.path.exists(self._evfile % mi)]:
                done.discard(mi)
                self.__dict__.get("_evals_full_mem", {}).pop(mi, None)
                self.__dict__.get("_extra_mem", {}).pop(mi, None)
        self.generate_ms(self.beamtransfer._my_ms(), regen)
        storage.flush()
        parallel.barrier()
        if parallel.rank0():
            logger.info("======== Ending KL calculation (time=%f) ========" % (time.time() - st))
        self._collect()
        if not storage.discard():
            # what this run remembered about its own m (spectra for `_collect`, the done set) ends with it: a later call
            # goes by the files alone
            for k in ("_done", "_evals_full_mem", "_extra_mem"):
                self.__dict__.pop(k, None)

    # ---- spectra ---------------------------------------------------------------------------
    def evals_all(self):
        with storage.File(self.evdir + "/evals.hdf5", "r") as f:
            return f["evals"][:]

    def _evfunc(self, mi):
        evf = np.zeros(self.beamtransfer.ndofmax)
        mem = self.__dict__.get("_evals_full_mem", {}).get(mi)
        if mem is not None:   # this process made the m: no need to open its file again
            if mem.size > 0:
                evf[-mem.size :] = mem
            return evf
        with storage.File(self._evfile % mi, "r") as f:
            if f["evals_full"].shape[0] > 0:
                ev = f["evals_full"][:]
                evf[-ev.size :] = ev
        return evf

    def _collect(self):
        """evals.hdf5: (mmax+1, ndofmax), right aligned (kltransform.py:452-478)."""
        mine = [(mi, self._evfunc(mi)) for mi in self.beamtransfer._my_ms()]
        parts = parallel.gather_objects(mine)
        if parallel.rank0():
            if os.path.exists(self.evdir + "/evals.hdf5") or storage.discard() or parallel.is_virtual():
                return
            arr = np.zeros((self.telescope.mmax + 1, self.beamtransfer.ndofmax))
            for part in parts:
                for mi, ev in part:
                    arr[mi] = ev
            with storage.File(self.evdir + "/evals.hdf5", "w") as f:
                f.create_dataset("evals", data=arr)

    # ---- mode access (kltransform.py:517-660) -------------------------------------------------
    olddatafile = False

    @util.cache_last
    def modes_m(self, mi, threshold=None, device=False):
        """``device=True`` (the Fisher estimators during generation): the eigenvectors of a batch in flight may come back
        as rows of the batch's DEVICE tensor; every other caller gets numpy arrays, as from the reference."""
        mc = self.__dict__.get("_mode_cache")
        if mc is not None and mi in mc:   # modes of the batch in flight (generate_ms), exactly what the file will hold
            evals, evecs = mc[mi]   # (evecs may be a DEVICE tensor here: rows of the batch's eigenvector matrix)
            if evals.shape[0] == 0:
                return None, None
            startind = np.searchsorted(evals, threshold) if threshold is not None else 0
            if startind == evals.size:
                return None, None
            evecs = evecs[startind:]
            if not device and not isinstance(evecs, np.ndarray):
                evecs = evecs.cpu().numpy()
            return evals[startind:], evecs
        if not os.path.exists(self._evfile % mi):
            return self.transform_save(mi)
        with storage.File(self._evfile % mi, "r") as f:
            if f["evals"].shape[0] == 0:
                return None, None
            evals = f["evals"][:]
            startind = np.searchsorted(evals, threshold) if threshold is not None else 0
            if startind == evals.size:
                return None, None
            evecs = f["evecs"][startind:]
            return evals[startind:], (evecs.conj() if self.olddatafile else evecs)

    @util.cache_last
    def evals_m(self, mi, threshold=None):
        modes = self.modes_m(mi, threshold)
        return None if modes[0] is None else modes[0]

    @util.cache_last
    def invmodes_m(self, mi, threshold=None):
        evals = self.evals_m(mi, threshold)
        with storage.File(self._evfile % mi, "r") as f:
            if "evinv" in f:
                inv = f["evinv"][:]
                if threshold is not None:
                    inv = inv[(-evals.size) :]
                return inv.T
        return np.linalg.pinv(self.modes_m(mi, threshold)[1])

    # ---- projections (kltransform.py:710-842) ---------------------------------------------------
    def project_vector_svd_to_kl(self, mi, vec, threshold=None):
        evals, evecs = self.modes_m(mi, threshold)
        if evals is None:
            return np.zeros((0,), dtype=np.complex128)
        if vec.shape[0] != evecs.shape[1]:
            raise Exception("Vectors are incompatible.")
        from .beamtransfer import _device_gemm

        return _device_gemm(evecs, vec)

    def project_vector_kl_to_svd(self, mi, vec, threshold=None):
        """Eigenbasis vector back into the SVD (telescope) basis through the stored inverse
        modes (kltransform.py:739-769)."""
        evals, evecs = self.modes_m(mi, threshold)
        if evals is None:
            return np.zeros(self.beamtransfer.ndofmax, dtype=np.complex128)
        if vec.shape[0] != evecs.shape[0]:
            raise Exception("Vectors are incompatible.")
        from .beamtransfer import _device_gemm

        return _device_gemm(self.invmodes_m(mi, threshold), vec)

    def skymodes_m(self, mi, threshold=None):
        """KL modes rotated onto the sky, [nmodes, nfreq, nsky] (kltransform.py:663-708; like the
        reference this assumes un-compressed modes of length nfreq * ntel)."""
        evals, evecs = self.modes_m(mi, threshold=threshold)
        if evals is None:
            raise Exception("Don't seem to be any evals to use.")
        bt = self.beamtransfer
        beam = bt.beam_m(mi).reshape((bt.nfreq, bt.ntel, bt.nsky))
        evecs = evecs.reshape((-1, bt.nfreq, bt.ntel))
        evsky = np.zeros((evecs.shape[0], bt.nfreq, bt.nsky), dtype=np.complex128)
        from .beamtransfer import _device_gemm

        for fi in range(bt.nfreq):
            evsky[:, fi, :] = _device_gemm(evecs[:, fi, :], beam[fi])
        return evsky

    def project_vector_sky_to_kl(self, mi, vec, threshold=None):
        return self.project_vector_svd_to_kl(mi, self.beamtransfer.project_vector_sky_to_svd(mi, vec), threshold)

    def project_matrix_svd_to_kl(self, mi, mat, threshold=None):
        evals, evecs = self.modes_m(mi, threshold)
        if (mat.shape[0] != evecs.shape[1]) or (mat.shape[0] != mat.shape[1]):
            raise Exception("Matrix size incompatible.")
        from .beamtransfer import _device_gemm

        return _device_gemm(_device_gemm(evecs, mat), evecs.T.conj())

    def project_matrix_sky_to_kl(self, mi, mat, threshold=None):
        return self.project_matrix_svd_to_kl(mi, self.beamtransfer.project_matrix_sky_to_svd(mi, mat), threshold)
