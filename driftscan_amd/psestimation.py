"""Quadratic power-spectrum estimation: exact Fisher matrix per m, GPU-backed.

Mirrors ``drift.core.psestimation.PSEstimation`` / ``PSExact`` (drift/core/psestimation.py:146-815):
same constructor, config properties, band bookkeeping, ``fisher_bias_m``, ``generate`` and the
datasets of ``fisher.hdf5``.  The per-m work — every band projected into the KL basis and the
weighted traces between all pairs of bands — runs on the GPU for batches of m-blocks
(``dm_fisher``); the sum over m is a sum over ranks' local m followed by one all-reduce (the only
collective of the path, psestimation.py:506-507).

The band angular power spectra ``clarray (nbands, lmax+1, nfreq, nfreq)`` are an INPUT: the
reference derives them from cora's 21cm correlation functions, which are not available here.  Set
``psobj.clarray`` to use your own; otherwise :func:`band_clarray_standin` cuts the built-in
analytic signal model (``skymodel.im21cm_model``) into multipole bands matched to the k bands in the
flat-sky sense (l = k_perp * chi) — documented as a stand-in, not as cora's bands.
"""
import logging
import os
import time

import numpy as np

from . import config, parallel, skymodel, storage
from .device import get_context

logger = logging.getLogger(__name__)


def range_config(lst):
    """Concatenated linspace/logspace segments (psestimation.py:58-87)."""
    lst2 = []
    endpoint = False
    count = 1
    for item in lst:
        if isinstance(item, dict):
            if count == len(lst):
                endpoint = True
            count += 1
            if item["spacing"] == "log":
                item = np.logspace(np.log10(item["start"]), np.log10(item["stop"]), item["num"], endpoint=endpoint)
            elif item["spacing"] == "linear":
                item = np.linspace(item["start"], item["stop"], item["num"], endpoint=endpoint)
            lst2.append(np.atleast_1d(item))
        else:
            raise Exception("Require a dict.")
    return np.concatenate(lst2)


def decorrelate_ps(ps, fisher):
    """Uncorrelated band powers through the Cholesky factor of the Fisher matrix (psestimation.py:90-121)."""
    fh = np.linalg.cholesky(fisher)
    fhi = np.linalg.inv(fh)
    m = fhi / fh.sum(axis=1)[:, np.newaxis]
    return np.dot(m, np.dot(fisher, ps))


_CHI_STANDIN = 3000.0  # Mpc/h, order of the comoving distance at z ~ 1-2: only used by the stand-in bands


def band_clarray_standin(lmax, frequencies, k_start, k_end):
    """Stand-in band spectra: band a = analytic signal model restricted to
    k_start[a] * chi <= l < k_end[a] * chi (see the module docstring)."""
    base = skymodel.im21cm_model(lmax, frequencies, 1)[0, 0]  # (L, F, F)
    ell = np.arange(lmax + 1)
    out = np.zeros((len(k_start),) + base.shape)
    for a, (ks, ke) in enumerate(zip(k_start, k_end)):
        sel = (ell >= ks * _CHI_STANDIN) & (ell < ke * _CHI_STANDIN)
        out[a, sel] = base[sel]
    return out


class PSEstimation(config.Reader):
    bandtype = config.Property(proptype=str, default="polar")
    k_bands = config.Property(proptype=range_config,
                              default=[{"spacing": "linear", "start": 0.0, "stop": 0.4, "num": 20}])
    num_theta = config.Property(proptype=int, default=1)
    kpar_bands = config.Property(proptype=range_config,
                                 default=[{"spacing": "linear", "start": 0.0, "stop": 0.4, "num": 20}])
    kperp_bands = config.Property(proptype=range_config,
                                  default=[{"spacing": "linear", "start": 0.0, "stop": 0.4, "num": 20}])
    threshold = config.Property(proptype=float, default=0.0)
    unit_bands = config.Property(proptype=config.truthy, default=True)
    zero_mean = config.Property(proptype=config.truthy, default=True)
    # MI355X-side knob: device memory budget for one batch of m-blocks
    ps_chunk_gb = config.Property(proptype=float, default=32.0)

    crosspower = False
    clarray = None
    fisher = None
    bias = None

    def __init__(self, kltrans, subdir="ps"):
        self.kltrans = kltrans
        self.telescope = kltrans.telescope
        self.psdir = self.kltrans.evdir + "/" + subdir + "/"
        if parallel.io_root() and not os.path.exists(self.psdir):
            os.makedirs(self.psdir)
        parallel.barrier()

    @property
    def nbands(self):
        return self.k_center.size

    def num_evals(self, mi):
        evals = self.kltrans.modes_m(mi, threshold=self.threshold, device=True)[0]
        return evals.size if evals is not None else 0

    # ---- bands (psestimation.py:256-349) ------------------------------------------------------
    def genbands(self):
        if self.bandtype == "polar":
            self.theta_bands = np.linspace(0.0, np.pi / 2.0, self.num_theta + 1, endpoint=True)
            kb, tb = np.broadcast_arrays(self.k_bands[np.newaxis, :], self.theta_bands[:, np.newaxis])
            self.k_start = kb[1:, :-1].flatten()
            self.k_end = kb[1:, 1:].flatten()
            self.k_center = 0.5 * (self.k_end + self.k_start)
            self.theta_start = tb[:-1, 1:].flatten()
            self.theta_end = tb[1:, 1:].flatten()
            self.theta_center = 0.5 * (self.theta_end + self.theta_start)
            lo, hi = self.k_start, self.k_end
        elif self.bandtype == "cartesian":
            kparb, kperpb = np.broadcast_arrays(self.kpar_bands[np.newaxis, :], self.kperp_bands[:, np.newaxis])
            self.kpar_start = kparb[1:, :-1].flatten()
            self.kpar_end = kparb[1:, 1:].flatten()
            self.kpar_center = 0.5 * (self.kpar_end + self.kpar_start)
            self.kperp_start = kperpb[:-1, 1:].flatten()
            self.kperp_end = kperpb[1:, 1:].flatten()
            self.kperp_center = 0.5 * (self.kperp_end + self.kperp_start)
            self.k_center = (self.kpar_center**2 + self.kperp_center**2) ** 0.5
            lo, hi = self.kperp_start, self.kperp_end
        else:
            raise Exception("Bandtype %s is not supported." % self.bandtype)
        self.band_power = np.ones_like(self.k_center)
        if self.clarray is None:
            logger.warning("PSEstimation: no band C_l supplied, using the analytic stand-in bands")
            self.clarray = band_clarray_standin(self.telescope.lmax, self.telescope.frequencies, lo, hi)
        if self.clarray.shape[0] != self.nbands:
            raise Exception("clarray has %d bands, the band configuration %d" % (self.clarray.shape[0], self.nbands))

    def delbands(self):
        self.clarray = None
        self.__dict__.pop("_cl_dev", None)

    # ---- per-m Fisher ----------------------------------------------------------------------------
    def fisher_bias_m(self, mi):
        """(fisher (nbands, nbands) complex, bias (nbands,)) of one m (psestimation.py:416-438)."""
        return self.fisher_bias_batch([mi])[0]

    def fisher_bias_batch(self, ms):
        raise NotImplementedError

    def _batches(self, ms):
        """Batches of m under the device budget: the vectorised band projections
        (nbands * nmodes^2) dominate."""
        bt = self.kltrans.beamtransfer
        budget = self.ps_chunk_gb * (1 << 30)
        cur, used = [], 0.0
        for mi in ms:
            n = float(bt.ndof(mi))
            need = 16.0 * (self.clarray.shape[0] * n * n + 4.0 * n * n)
            if cur and used + need > budget:
                yield cur
                cur, used = [], 0.0
            cur.append(mi)
            used += need
        if cur:
            yield cur

    def accumulate_ms(self, ms):
        """Add the Fisher matrix / bias of the given m (of this rank) to the local sums: the per-batch step that
        `ProductManager.generate` runs right behind the KL transform of those m, while their SVD products are resident
        and their modes are in the KL object's mode cache.  `generate` finishes the rest and does the all-reduce."""
        if self.clarray is None:
            self.genbands()
        acc = self.__dict__.setdefault("_acc", dict(fisher=np.zeros((self.nbands, self.nbands)), bias=np.zeros(self.nbands),
                                                    done=set()))
        todo = [mi for mi in ms if mi not in acc["done"]]
        for batch in self._batches(todo):
            for f, b in self.fisher_bias_batch(batch):
                acc["fisher"] += f.real
                acc["bias"] += b.real
            acc["done"].update(batch)

    # ---- total Fisher (psestimation.py:463-560) ----------------------------------------------------
    def generate(self, regen=False):
        st = time.time()
        ffile = self.psdir + "/fisher.hdf5"
        if storage.can_open(ffile) and not regen:
            logger.info("Fisher matrix file: %s exists. Skipping..." % ffile)
            return
        parallel.barrier()
        self.genbands()
        acc = self.__dict__.pop("_acc", None)
        fisher_loc = np.zeros((self.nbands, self.nbands), dtype=np.float64)
        bias_loc = np.zeros(self.nbands, dtype=np.float64)
        if acc is not None:
            # the resident pipeline has done (some of) this rank's own m already: keep to its partition
            fisher_loc += acc["fisher"]
            bias_loc += acc["bias"]
            ms = [mi for mi in self.kltrans.beamtransfer._my_ms() if mi not in acc["done"]]
        else:
            ms = parallel.partition(list(range(self.telescope.mmax + 1)),
                                    costs=[float(self.kltrans.beamtransfer.ndof(mi)) ** 3 + 1.0 for mi in
                                           range(self.telescope.mmax + 1)])
        for batch in self._batches(ms):
            for f, b in self.fisher_bias_batch(batch):
                fisher_loc += f.real  # "be careful of the .real here" (psestimation.py:497-502)
                bias_loc += b.real
        self.fisher = parallel.allreduce_sum(fisher_loc)
        self.bias = parallel.allreduce_sum(bias_loc)
        if parallel.rank0():
            logger.info("======== Ending PS calculation (time=%f) ========" % (time.time() - st))
            if not (self.fisher == 0).all():
                import scipy.linalg as la

                cv = la.pinv(self.fisher, atol=1e-8)
                err = cv.diagonal() ** 0.5
                cr = cv / np.outer(err, err)
            else:
                cv = np.zeros_like(self.fisher)
                err = cv.diagonal()
                cr = np.zeros_like(self.fisher)
            if storage.discard() or parallel.is_virtual():   # (an emulated share holds one rank's part of the sum only)
                parallel.barrier()
                return
            with storage.File(ffile, "w") as f:
                f.attrs["bandtype"] = np.bytes_(self.bandtype)
                f.create_dataset("fisher", data=self.fisher)
                f.create_dataset("bias", data=self.bias)
                f.create_dataset("covariance", data=cv)
                f.create_dataset("errors", data=err)
                f.create_dataset("correlation", data=cr)
                f.create_dataset("band_power", data=self.band_power)
                if self.bandtype == "polar":
                    for k in ("k_start", "k_end", "k_center", "theta_start", "theta_end", "theta_center", "k_bands",
                              "theta_bands"):
                        f.create_dataset(k, data=getattr(self, k))
                else:
                    for k in ("kpar_start", "kpar_end", "kpar_center", "kperp_start", "kperp_end", "kperp_center",
                              "kpar_bands", "kperp_bands"):
                        f.create_dataset(k, data=getattr(self, k))
        parallel.barrier()

    def fisher_file(self):
        return storage.File(self.psdir + "/fisher.hdf5", "r")

    def fisher_bias(self):
        with storage.File(self.psdir + "/fisher.hdf5", "r") as f:
            return f["fisher"][:], f["bias"][:]


def _linear_offsets(sizes):
    off = np.concatenate([[0], np.cumsum(np.asarray(sizes, dtype=np.int64))])
    return off[:-1].copy(), int(off[-1])


class PSExact(PSEstimation):
    """Exact Fisher matrix by forward projection of every band (psestimation.py:657-815)."""

    def fisher_bias_batch(self, ms):
        """[(fisher, bias)] for the given m, one dm_fisher call for the whole batch."""
        if self.clarray is None:
            self.genbands()
        ctx = get_context()
        kl = self.kltrans
        bt = kl.beamtransfer
        nb = self.clarray.shape[0]
        zero = (np.zeros((nb, nb), dtype=np.complex128), np.zeros(nb, dtype=np.complex128))
        modes = [kl.modes_m(mi, threshold=self.threshold, device=True) for mi in ms]
        nmodes = np.array([0 if ev is None else ev.size for ev, _ in modes], dtype=np.int64)
        if nmodes.sum() == 0:
            return [zero for _ in ms]
        import torch

        bsvd = bt._stacked_products(ms, "beam_svd")
        svnum = np.stack([bt._svd_num(mi)[0] for mi in ms])
        ndofs = svnum.sum(axis=1)
        eoff, etot = _linear_offsets(nmodes * ndofs)
        voff, vtot = _linear_offsets(nmodes)
        Vh = np.zeros(max(vtot, 1), dtype=np.float64)
        parts = []   # the modes back to back on the device: rows of the batch's eigenvector matrix when they come from the
        for i, (ev, E) in enumerate(modes):   # KL object's mode cache (generate_ms), host arrays when they come from a file
            if nmodes[i] == 0:
                continue
            if E.shape[1] != ndofs[i]:
                raise Exception("KL modes of m=%d have length %d, the SVD basis %d" % (ms[i], E.shape[1], ndofs[i]))
            parts.append(ctx.to_device(np.ascontiguousarray(E).ravel()) if isinstance(E, np.ndarray) else E.reshape(-1))
            Vh[voff[i] : voff[i] + nmodes[i]] = ev
        Ed = torch.cat(parts) if len(parts) > 1 else parts[0].contiguous()
        # (nbands, L, F, F) -> (nbands, F, F, L): the contraction index innermost, as dm_project_cov reads it
        cache = self.__dict__.get("_cl_dev")
        if cache is None or cache[0] is not self.clarray or cache[1].device.index != ctx.device:
            # uploaded as it lies; the (.., F, F, L) layout and the symmetry test are made on the device (on the host the
            # strided transposition and comparison of the configs[3] table — 151 MB — are a second of idle GPU per rank)
            dev0 = ctx.to_device(np.ascontiguousarray(np.asarray(self.clarray, dtype=np.float64)))
            cache = (self.clarray, dev0.permute(0, 2, 3, 1).contiguous(),
                     bool(torch.equal(dev0, dev0.transpose(2, 3))))  # f <-> f' symmetry, checked not assumed
            del dev0
            self.__dict__["_cl_dev"] = cache  # the band tables do not change between batches (151 MB at config 3)
        cl = cache[1]
        F = ctx.fisher(bsvd, svnum, np.array(ms), cl, Ed, eoff, nmodes, ctx.to_device(Vh), voff,
                       cl_symmetric=cache[2])
        Fh = F.cpu().numpy()
        return [(Fh[i], np.zeros(nb, dtype=np.complex128)) if nmodes[i] > 0 else zero for i in range(len(ms))]

    def _work_fisher_bias_m(self, mi):
        return self.fisher_bias_m(mi)
