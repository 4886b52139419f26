"""Two-stage KL foreground filter, GPU-backed (drift/core/doublekl.py:16-128).

Stage 1 diagonalises signal against foregrounds (thermal noise switched off down to a
1 mK floor) and keeps the modes with S/F above ``foreground_threshold``; stage 2
diagonalises signal against foregrounds + noise inside that subspace.  The
re-projections E C E^H (doublekl.py:73-74) and the mode composition E2^H E (:80) are
grouped ZGEMMs; both eigenproblems go through ``dm_eigh_gen``.
"""
import os

import numpy as np

from . import config, kltransform, parallel, storage
from ._lib import block_offsets
from .device import get_context


class DoubleKL(kltransform.KLTransform):
    foreground_threshold = config.Property(proptype=float, default=100.0)

    def _transform_batch(self, ms, to_host=True):
        """Two-stage KL of several m at once (doublekl.py:30-87).  Every product of stage 2 — E C E^H for the
        signal and noise matrices, the composition E2 E, the inverses — is ONE grouped launch over all live
        blocks; nothing is copied to the host until the end."""
        import torch

        ctx = get_context()
        nb = len(ms)
        thr = float(self.foreground_threshold)
        # ---- stage 1: S vs F (use_thermal = False, doublekl.py:44-52)
        self.use_thermal = False
        try:
            S, N, ndofs, off = self.sn_covariance_device(ms)
        finally:
            self.use_thermal = True
        ndofs = [int(n) for n in ndofs]
        # Stage 2 needs the same S, and N with the thermal term at full strength instead of the 1 mK floor
        # (kltransform.py:292-303): N2 = N1 + (1 - nc1) U diag(noise) U^H.  eigh_gen destroys its inputs,
        # so keep copies instead of projecting both sky covariances a second time.
        S2, N2 = S.clone(), N.clone()
        N1 = N.clone() if self.inverse else None   # inv_gen(E1) = N1 E1^H needs the stage-1 noise matrix itself
        # only the modes with S/F strictly above the threshold are used below (doublekl.py:56-60): the others
        # are not back-transformed unless the inverse needs the whole of E1
        cut1 = None if self.inverse else ("upper", float(np.nextafter(thr, np.inf)))
        ev1, evoff1, E1, ac1, _ = ctx.eigh_gen(S, N, ndofs, off, cut=cut1)
        del S, N
        ev1_h = ev1.cpu().numpy()
        f_evals = [ev1_h[evoff1[i] : evoff1[i] + ndofs[i]].copy() for i in range(nb)]
        # eigenvalues ascend, so the modes with S/F above the threshold are the trailing rows of E1
        keep = [int((fe > thr).sum()) for fe in f_evals]
        live = [i for i in range(nb) if keep[i] > 0]
        results = [None] * nb
        for i in range(nb):
            if ndofs[i] == 0:
                results[i] = (np.array([]), np.array([[]]), np.array([[]]), {"ac": 0.0, "f_evals": np.array([])})
            elif keep[i] == 0:
                z = np.zeros((0, ndofs[i]), dtype=np.complex128)
                results[i] = (np.array([]), z, z.copy() if self.inverse else None,
                              {"ac": float(ac1[i]), "f_evals": f_evals[i]})
        if not live:
            return results

        def rows_kept(i):  # kept rows of E1 (r x n)
            n, r = ndofs[i], keep[i]
            return E1[off[i] + (n - r) * n : off[i] + n * n]

        # ---- stage 2 covariances (doublekl.py:70-74)
        nc1 = (1e-3 / self.telescope.tsys_flat) ** 2
        bt = self.beamtransfer
        but = bt._stacked_products(ms, "beam_ut")
        svnum = np.stack([bt._svd_num(mi)[0] for mi in ms])
        if self.use_thermal:  # always true here; spelled out to mirror sn_covariance
            ctx.project_diag(but, svnum, self._npower_device(1.0), N2, off, alpha=1.0 - nc1, accumulate=True)
        n2 = np.array([keep[i] for i in live], dtype=np.int64)
        off2, tot2 = block_offsets(n2)
        rn = np.array([keep[i] * ndofs[i] for i in live], dtype=np.int64)
        toff = np.concatenate([[0], np.cumsum(rn)])
        cs = ctx.empty((max(tot2, 1),), np.complex128)
        cn = ctx.empty((max(tot2, 1),), np.complex128)
        tmp = ctx.empty((2, max(int(toff[-1]), 1)), np.complex128)
        p1, p2 = [], []
        for k, i in enumerate(live):
            n, r = ndofs[i], keep[i]
            Ei = rows_kept(i)
            for j, (src, dst) in enumerate(((S2, cs), (N2, cn))):
                Ti = tmp[j, toff[k] : toff[k] + r * n]
                p1.append(dict(A=Ei, B=src[off[i] : off[i] + n * n], C=Ti, M=r, N=n, K=n, rsA=n, csA=1, rsB=n, csB=1, ldc=n))
                p2.append(dict(A=Ti, B=Ei, C=dst[off2[k] : off2[k] + r * r], M=r, N=r, K=n, rsA=n, csA=1, rsB=1, csB=n,
                               conjB=True, ldc=r))
        ctx.zgemm_grouped(p1)   # E C
        ctx.zgemm_grouped(p2)   # (E C) E^H
        cn_keep = cn.clone() if self.inverse else None
        # stage-2 modes below the S/N threshold are dropped by transform_save (kltransform.py:388-398)
        cut2 = ("upper", self.threshold) if (self.subset and not self.inverse) else None
        ev2, evoff2, E2, ac2, _ = ctx.eigh_gen(cs, cn, n2, off2, cut=cut2)
        # ---- modes = E2 . E1[kept] (doublekl.py:80)
        modes = ctx.empty((max(int(toff[-1]), 1),), np.complex128)
        ctx.zgemm_grouped([dict(A=E2[off2[k] : off2[k] + keep[i] ** 2], B=rows_kept(i),
                                C=modes[toff[k] : toff[k] + keep[i] * ndofs[i]], M=keep[i], N=ndofs[i], K=keep[i],
                                rsA=keep[i], csA=1, rsB=ndofs[i], csB=1, ldc=ndofs[i]) for k, i in enumerate(live)])
        inv = None
        if self.inverse:
            # doublekl.py:63-67, :83-85: inv = inv_gen(evecs2) . inv_gen(E1).T[ind].  With E N E^H = I the inverse of
            # a mode matrix is N E^H, so inv_gen(E1).T = conj(E1 N1) and inv_gen(evecs2) = inv(E2^H) = E2 N2'.
            for k, i in enumerate(live):   # the non-positive-definite rescue shifted diag(N) (kltransform.py:101-111)
                if ac1[i] != 0.0:
                    N1[off[i] : off[i] + ndofs[i] ** 2].view(ndofs[i], ndofs[i]).diagonal().add_(float(ac1[i]))
                if ac2[k] != 0.0:
                    cn_keep[off2[k] : off2[k] + keep[i] ** 2].view(keep[i], keep[i]).diagonal().add_(float(ac2[k]))
            inv1 = tmp[0]
            inv2 = ctx.empty((max(tot2, 1),), np.complex128)
            q1, q2, q3 = [], [], []
            inv = ctx.empty((max(int(toff[-1]), 1),), np.complex128)
            for k, i in enumerate(live):
                n, r = ndofs[i], keep[i]
                q1.append(dict(A=rows_kept(i), B=N1[off[i] : off[i] + n * n], C=inv1[toff[k] : toff[k] + r * n], M=r, N=n, K=n,
                               rsA=n, csA=1, rsB=n, csB=1, ldc=n, conjA=True, conjB=True))
                q2.append(dict(A=E2[off2[k] : off2[k] + r * r], B=cn_keep[off2[k] : off2[k] + r * r],
                               C=inv2[off2[k] : off2[k] + r * r], M=r, N=r, K=r, rsA=r, csA=1, rsB=r, csB=1, ldc=r))
                q3.append(dict(A=inv2[off2[k] : off2[k] + r * r], B=inv1[toff[k] : toff[k] + r * n],
                               C=inv[toff[k] : toff[k] + r * n], M=r, N=n, K=r, rsA=r, csA=1, rsB=n, csB=1, ldc=n))
            ctx.zgemm_grouped(q1 + q2)
            ctx.zgemm_grouped(q3)
        if not to_host:
            for k, i in enumerate(live):
                n, r = ndofs[i], keep[i]
                results[i] = (ev2[evoff2[k] : evoff2[k] + r], modes[toff[k] : toff[k] + r * n].view(r, n),
                              None if inv is None else inv[toff[k] : toff[k] + r * n].view(r, n),
                              {"ac": float(ac1[i]), "f_evals": f_evals[i]})
            return results
        ev2_h = ev2.cpu().numpy()
        modes_h = ctx.to_host(modes)
        inv_h = ctx.to_host(inv) if inv is not None else None
        for k, i in enumerate(live):
            n, r = ndofs[i], keep[i]
            results[i] = (ev2_h[evoff2[k] : evoff2[k] + r].copy(), modes_h[toff[k] : toff[k] + r * n].reshape(r, n),
                          None if inv_h is None else inv_h[toff[k] : toff[k] + r * n].reshape(r, n),
                          {"ac": float(ac1[i]), "f_evals": f_evals[i]})   # `ac` is the stage-1 value (doublekl.py:58)
        return results

    def _ev_save_hook(self, f, evextra):
        kltransform.KLTransform._ev_save_hook(self, f, evextra)
        f.create_dataset("f_evals", data=evextra["f_evals"])

    def _collect(self):
        """evals.hdf5 with both spectra (doublekl.py:95-128)."""
        nd = self.beamtransfer.ndofmax

        def evfunc(mi):
            ta = np.zeros((2, nd))
            mem = self.__dict__.get("_evals_full_mem", {}).get(mi)
            if mem is not None:   # this process made the m: no need to open its file again
                fev = self.__dict__["_extra_mem"][mi]["f_evals"]
                fev = np.asarray(fev.cpu().numpy() if hasattr(fev, "cpu") else fev)
                if mem.size > 0:
                    ta[0, -mem.size :] = mem
                    ta[1, -fev.size :] = fev
                return ta
            with storage.File(self._evfile % mi, "r") as f:
                if f["evals_full"].shape[0] > 0:
                    ev, fev = f["evals_full"][:], f["f_evals"][:]
                    ta[0, -ev.size :] = ev
                    ta[1, -fev.size :] = fev
            return ta

        mine = [(mi, evfunc(mi)) for mi in self.beamtransfer._my_ms()]
        parts = parallel.gather_objects(mine)
        if parallel.rank0():
            fname = self.evdir + "/evals.hdf5"
            if os.path.exists(fname) or storage.discard() or parallel.is_virtual():
                return
            arr = np.zeros((self.telescope.mmax + 1, 2, nd))
            for part in parts:
                for mi, ta in part:
                    arr[mi] = ta
            with storage.File(fname, "w") as f:
                f.create_dataset("evals", data=arr[:, 0])
                f.create_dataset("f_evals", data=arr[:, 1])
