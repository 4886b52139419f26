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

    def _transform_batch(self, ms):
        ctx = get_context()
        nb = len(ms)
        # ---- stage 1: S vs F (use_thermal = False, doublekl.py:44-52)
        self.use_thermal = False
        S, N, ndofs, off = self.sn_covariance_device(ms)
        # only the modes with S/F strictly above the threshold are used below (doublekl.py:56-60): the others
        # are not back-transformed (their rows of E1 stay zero); all eigenvalues are returned either way
        ev1, evoff1, E1, ac1, _ = ctx.eigh_gen(S, N, ndofs, off,
                                               cut=("upper", float(np.nextafter(self.foreground_threshold, np.inf))))
        ev1_h = ev1.cpu().numpy()
        f_evals = [ev1_h[evoff1[i] : evoff1[i] + int(ndofs[i])].copy() for i in range(nb)]
        # modes with S/F above the threshold: eigenvalues ascend, so they are the trailing rows
        keep = [int((fe > self.foreground_threshold).sum()) for fe in f_evals]
        results = [None] * nb
        live = [i for i in range(nb) if keep[i] > 0]
        for i in range(nb):
            if ndofs[i] == 0:
                results[i] = (np.array([]), np.array([[]]), np.array([[]]), {"ac": 0.0, "f_evals": np.array([])})
            elif keep[i] == 0:
                n = int(ndofs[i])
                results[i] = (np.array([]), np.zeros((0, n), dtype=np.complex128), None,
                              {"ac": float(ac1[i]), "f_evals": f_evals[i]})
        if live:
            # ---- stage 2: full S, N projected into the kept subspace (doublekl.py:70-80)
            self.use_thermal = True
            S2, N2, _, _ = self.sn_covariance_device(ms)
            n2 = np.array([keep[i] for i in live], dtype=np.int64)
            off2, tot2 = block_offsets(n2)
            cs = ctx.empty((max(tot2, 1),), np.complex128)
            cn = ctx.empty((max(tot2, 1),), np.complex128)
            tmp_off = np.concatenate([[0], np.cumsum([keep[i] * int(ndofs[i]) for i in live])])
            tmp = ctx.empty((max(int(tmp_off[-1]), 1),), np.complex128)
            for src, dst in ((S2, cs), (N2, cn)):
                for k, i in enumerate(live):
                    n, r = int(ndofs[i]), keep[i]
                    Ei = E1[off[i] + (n - r) * n : off[i] + n * n]          # kept rows of E (r x n)
                    Ci = src[off[i] : off[i] + n * n]
                    Ti = tmp[tmp_off[k] : tmp_off[k] + r * n]
                    ctx.zgemm(Ei, Ci, Ti, r, n, n, rsA=n, csA=1, rsB=n, csB=1, ldc=n)               # E C
                    ctx.zgemm(Ti, Ei, dst[off2[k] : off2[k] + r * r], r, r, n, rsA=n, csA=1, rsB=1, csB=n,
                              conjB=True, ldc=r)                                                     # (E C) E^H
            ev2, evoff2, E2, ac2, _ = ctx.eigh_gen(cs, cn, n2, off2)
            ev2_h = ev2.cpu().numpy()
            for k, i in enumerate(live):
                n, r = int(ndofs[i]), keep[i]
                Ei = E1[off[i] + (n - r) * n : off[i] + n * n]
                out = ctx.empty((r * n,), np.complex128)
                # rows of E2 are the stage-2 modes in the stage-1 basis: modes = E2 . E
                ctx.zgemm(E2[off2[k] : off2[k] + r * r], Ei, out, r, n, r, rsA=r, csA=1, rsB=n, csB=1, ldc=n)
                ctx.sync()
                evecs = out.cpu().numpy().reshape(r, n)
                inv = None
                if self.inverse:
                    inv = kltransform._inv_gen(evecs).T
                results[i] = (ev2_h[evoff2[k] : evoff2[k] + r].copy(), evecs, inv,
                              {"ac": float(ac2[k]), "f_evals": f_evals[i]})
        return results

    def _ev_save_hook(self, f, evextra):
        kltransform.KLTransform._ev_save_hook(self, f, evextra)
        f.create_dataset("f_evals", data=evextra["f_evals"])

    def _collect(self):
        """evals.hdf5 with both spectra (doublekl.py:95-128)."""
        nd = self.beamtransfer.ndofmax

        def evfunc(mi):
            ta = np.zeros((2, nd))
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
            if os.path.exists(fname):
                return
            arr = np.zeros((self.telescope.mmax + 1, 2, nd))
            for part in parts:
                for mi, ta in part:
                    arr[mi] = ta
            with storage.File(fname, "w") as f:
                f.create_dataset("evals", data=arr[:, 0])
                f.create_dataset("f_evals", data=arr[:, 1])
