"""Timestream simulation, m-mode transform and map-making on top of the GPU-backed operators.

Mirrors ``drift.pipeline.timestream`` (drift/pipeline/timestream.py:15-829): the ``Timestream`` class with the
reference's file layout (``timestream_f/<f>/timestream.hdf5``, ``mmodes/<m>/mode.hdf5``, ``svd.hdf5``,
``klmode_<name>_<thr>.hdf5``, ``klmodes_<name>_<thr>.hdf5``), ``simulate`` and the three map-makers.  These are
CONSUMERS of the per-m operators: every projection goes through
``BeamTransfer`` / ``KLTransform`` (grouped ZGEMMs on the device); the sky-map <-> a_lm transforms
(``cora.util.hputil.sphtrans_sky`` / ``sphtrans_inv_sky`` in the reference, not available here) are the
``healpix`` restatements of this package.  Ranks split frequencies and m as the reference does, but without its two
MPI transposes: timestream files are per frequency and m-mode files per m, so a rank reads the frequencies it needs.
"""
import os
import pickle

import numpy as np

from . import healpix, parallel, storage, util


class Timestream(object):
    directory = None
    output_directory = None
    beamtransfer_dir = None
    no_m_zero = True

    def __init__(self, tsdir, prodmanager):
        self.directory = os.path.abspath(tsdir)
        self.output_directory = self.directory
        self.manager = prodmanager

    # ---- products this timestream belongs to (timestream.py:41-60) -------------------------------
    @property
    def beamtransfer(self):
        return self.manager.beamtransfer

    @property
    def telescope(self):
        return self.beamtransfer.telescope

    # ---- frequency-ordered files (timestream.py:63-100) ------------------------------------------------
    def _fdir(self, fi):
        return (self.directory + "/timestream_f/" + util.natpattern(self.telescope.nfreq)) % fi

    def _ffile(self, fi):
        return self._fdir(fi) + "/timestream.hdf5"

    @property
    def ntime(self):
        with storage.File(self._ffile(0), "r") as f:
            return int(f.attrs["ntime"])

    def timestream_f(self, fi):
        """[npairs, ntime] visibility timestream of one frequency."""
        with storage.File(self._ffile(fi), "r") as f:
            return f["timestream"][:]

    # ---- m-modes (timestream.py:103-185) ------------------------------------------------------------------
    def _mdir(self, mi):
        return (self.output_directory + "/mmodes/" + util.natpattern(self.telescope.mmax)) % abs(mi)

    def _mfile(self, mi):
        return self._mdir(mi) + "/mode.hdf5"

    def mmode(self, mi):
        """[nfreq, 2, npairs] visibility m-mode."""
        with storage.File(self._mfile(mi), "r") as f:
            return f["mmode"][:]

    def generate_mmodes(self):
        """FFT the timestreams along time and regroup by m: +m in slot 0, the conjugate of -m in slot 1."""
        marker = self.output_directory + "/mmodes/COMPLETED_M"
        if os.path.exists(marker):
            return
        tel = self.telescope
        mmax, nfreq, ntime = tel.mmax, tel.nfreq, self.ntime
        # Each rank transforms ITS frequencies, the result is regrouped by m across ranks (the reference's MPI transpose,
        # timestream.py:150-170): no rank reads every timestream file or holds the full (nfreq, 2, npairs, mmax + 1) array.
        nranks = parallel.size() if parallel._dist() else 1
        all_m = list(range(mmax + 1))
        m_of = [parallel.partition_for(all_m, r, nranks) for r in range(nranks)]
        f_of = [parallel.partition_for(list(range(nfreq)), r, nranks) for r in range(nranks)]
        me = parallel.rank() if parallel._dist() else 0
        local_f = f_of[me]
        pairs = np.zeros((len(local_f), 2, tel.npairs, mmax + 1), dtype=np.complex128)
        for k, fi in enumerate(local_f):
            row = np.fft.fft(self.timestream_f(fi), axis=-1) / ntime        # (npairs, ntime)
            pairs[k, 0, :, 0] = row[:, 0]
            pairs[k, 0, :, 1:] = row[:, 1 : mmax + 1]
            pairs[k, 1, :, 1:] = row[:, : -mmax - 1 : -1].conj()
        got = parallel.exchange([np.ascontiguousarray(pairs[..., m_of[r]]) for r in range(nranks)])
        mine = m_of[me]
        if mine:
            full = np.zeros((nfreq, 2, tel.npairs, len(mine)), dtype=np.complex128)
            for src, part in enumerate(got):
                if len(f_of[src]):
                    full[f_of[src]] = part
            for k, mi in enumerate(mine):
                os.makedirs(self._mdir(mi), exist_ok=True)
                with storage.File(self._mfile(mi), "w") as f:
                    f.create_dataset("mmode", data=np.ascontiguousarray(full[..., k]))
                    f.attrs["m"] = mi
        parallel.barrier()
        if parallel.rank0():
            open(marker, "a").close()
        parallel.barrier()

    # ---- SVD m-modes (timestream.py:191-231) ----------------------------------------------------------------
    def _svdfile(self, mi):
        return self._mdir(mi) + "/svd.hdf5"

    def mmode_svd(self, mi):
        with storage.File(self._svdfile(mi), "r") as f:
            if f["mmode_svd"].shape[0] == 0:
                return np.zeros((0,), dtype=np.complex128)
            return f["mmode_svd"][:]

    def generate_mmodes_svd(self):
        tel = self.telescope
        for mi in parallel.partition(list(range(tel.mmax + 1))):
            if os.path.exists(self._svdfile(mi)):
                continue
            tm = self.mmode(mi).reshape(tel.nfreq, 2 * tel.npairs)
            svdm = self.beamtransfer.project_vector_telescope_to_svd(mi, tm)
            with storage.File(self._svdfile(mi), "w") as f:
                f.create_dataset("mmode_svd", data=svdm)
                f.attrs["m"] = mi
        parallel.barrier()

    # ---- map-making (timestream.py:237-300, :400-457) -----------------------------------------------------
    def _alm_to_map(self, make_alm, nside, mapname, mlist=None):
        tel = self.telescope
        mlist = list(range(tel.mmax + 1)) if mlist is None else mlist
        mine = parallel.partition(mlist)
        parts = parallel.gather_objects([(mi, make_alm(mi)) for mi in mine])
        if parallel.rank0():
            alm = np.zeros((tel.nfreq, tel.num_pol_sky, tel.lmax + 1, tel.lmax + 1), dtype=np.complex128)
            for part in parts:
                for mi, a in part:
                    alm[..., mi] = a
            skymap = healpix.sphtrans_inv_sky(alm, nside)
            with storage.File(self.output_directory + "/" + mapname, "w") as f:
                f.create_dataset("map", data=skymap)
        parallel.barrier()

    def mapmake_full(self, nside, mapname):
        self._alm_to_map(lambda mi: self.beamtransfer.project_vector_telescope_to_sky(mi, self.mmode(mi)), nside, mapname)

    def mapmake_svd(self, nside, mapname):
        self.generate_mmodes_svd()
        self._alm_to_map(lambda mi: self.beamtransfer.project_vector_svd_to_sky(mi, self.mmode_svd(mi)), nside, mapname)

    # ---- KL m-modes (timestream.py:306-396) ---------------------------------------------------------------
    def set_kltransform(self, klname, threshold=None):
        self.klname = klname
        if threshold is None:
            threshold = self.manager.kltransforms[self.klname].threshold
        self.klthreshold = threshold

    def _klfile(self, mi):
        return self._mdir(mi) + ("/klmode_%s_%f.hdf5" % (self.klname, self.klthreshold))

    def mmode_kl(self, mi):
        with storage.File(self._klfile(mi), "r") as f:
            if f["mmode_kl"].shape[0] == 0:
                return np.zeros((0,), dtype=np.complex128)
            return f["mmode_kl"][:]

    def generate_mmodes_kl(self):
        kl = self.manager.kltransforms[self.klname]
        for mi in parallel.partition(list(range(self.telescope.mmax + 1))):
            if os.path.exists(self._klfile(mi)):
                continue
            klm = kl.project_vector_svd_to_kl(mi, self.mmode_svd(mi), threshold=self.klthreshold)
            with storage.File(self._klfile(mi), "w") as f:
                f.create_dataset("mmode_kl", data=klm)
                f.attrs["m"] = mi
        parallel.barrier()

    def collect_mmodes_kl(self):
        nd = self.beamtransfer.ndofmax

        def evfunc(mi):
            evf = np.zeros(nd, dtype=np.complex128)
            ev = self.mmode_kl(mi)
            if ev.size > 0:
                evf[-ev.size:] = ev
            return evf

        mine = [(mi, evfunc(mi)) for mi in parallel.partition(list(range(self.telescope.mmax + 1)))]
        parts = parallel.gather_objects(mine)
        if parallel.rank0():
            fname = self.output_directory + ("/klmodes_%s_%f.hdf5" % (self.klname, self.klthreshold))
            if os.path.exists(fname):
                return
            arr = np.zeros((self.telescope.mmax + 1, nd), dtype=np.complex128)
            for part in parts:
                for mi, ev in part:
                    arr[mi] = ev
            with storage.File(fname, "w") as f:
                f.create_dataset("evals", data=arr)

    def fake_kl_data(self):
        kl = self.manager.kltransforms[self.klname]
        for mi in parallel.partition(list(range(self.telescope.mmax + 1))):
            evals = kl.evals_m(mi)
            if evals is None:
                klmode = np.array([], dtype=np.complex128)
            else:
                modeamp = ((evals + 1.0) / 2.0) ** 0.5
                klmode = modeamp * (np.array([1.0, 1.0j]) * np.random.standard_normal((modeamp.shape[0], 2))).sum(axis=1)
            with storage.File(self._klfile(mi), "w") as f:
                f.create_dataset("mmode_kl", data=klmode)
                f.attrs["m"] = mi
        parallel.barrier()

    def mapmake_kl(self, nside, mapname, wiener=False):
        mapfile = self.output_directory + "/" + mapname
        if os.path.exists(mapfile):
            return
        kl = self.manager.kltransforms[self.klname]
        if not kl.inverse:
            raise Exception("Need the inverse to make a meaningful map.")

        def make_alm(mi):
            klmode = self.mmode_kl(mi)
            if klmode.size == 0:
                tel = self.telescope
                return np.zeros((tel.nfreq, tel.num_pol_sky, tel.lmax + 1), dtype=np.complex128)
            if wiener:
                evals = kl.evals_m(mi, self.klthreshold)
                if evals is not None:
                    klmode = klmode * (evals / (1.0 + evals))
            isvdmode = kl.project_vector_kl_to_svd(mi, klmode, threshold=self.klthreshold)
            return self.beamtransfer.project_vector_svd_to_sky(mi, isvdmode)

        mlist = list(range(1 if self.no_m_zero else 0, self.telescope.mmax + 1))
        self._alm_to_map(make_alm, nside, mapname, mlist=mlist)

    # ---- persistence (timestream.py:525-566) -----------------------------------------------------------------
    def __getstate__(self):
        # The reference pickles its ProductManager along with the object; the manager here owns device buffers, so the
        # pickle carries the directory of the products instead and `load` re-opens them (config.yaml is written there
        # by ProductManager.from_config, manager.py:134-162).
        state = {k: v for k, v in self.__dict__.items() if not k.startswith("_") and k != "manager"}
        state["manager_directory"] = getattr(self.manager, "directory", self.manager)
        return state

    def __setstate__(self, state):
        from . import manager as _manager

        mdir = state.pop("manager_directory", None)
        self.__dict__.update(state)
        self.manager = _manager.ProductManager.from_config(mdir) if isinstance(mdir, str) else mdir

    @property
    def _picklefile(self):
        return self.output_directory + "/timestreamobject.pickle"

    def save(self):
        if parallel.rank0():
            with open(self._picklefile, "wb") as f:
                pickle.dump(self, f)

    @classmethod
    def load(cls, tsdir):
        with open(cls(tsdir, tsdir)._picklefile, "rb") as f:
            return pickle.load(f)


def simulate(m, outdir, maps=(), ndays=None, resolution=0, seed=None, **kwargs):
    """Simulated timestream of the telescope of ProductManager `m` (timestream.py:645-829).

    maps: list of files holding a dataset `map` [freq, pol, pixel] whose sum is the sky; ndays = 0: no noise;
    resolution = 0: 2 mmax + 1 time samples.  Returns the Timestream."""
    bt = m.beamtransfer
    tel = bt.telescope
    lmax, mmax, nfreq, npol = tel.lmax, tel.mmax, tel.nfreq, tel.num_pol_sky
    if ndays is None:
        ndays = tel.ndays
    ntime = 2 * mmax + 1 if resolution == 0 else int(np.round(24 * 3600.0 / resolution))
    nranks = parallel.size() if parallel._dist() else 1
    me = parallel.rank() if parallel._dist() else 0
    local_freq = parallel.partition_for(list(range(nfreq)), me, nranks)
    lfreq = len(local_freq)
    col_vis = np.zeros((tel.npairs, lfreq, ntime), dtype=np.complex128)

    if len(maps) > 0:
        # The reference's two MPI transposes (timestream.py:700-760): every rank transforms the maps of ITS frequencies,
        # the a_lm are regrouped by m, every rank projects ITS m through the beam (all frequencies of an m in one grouped
        # product on the device), and the visibilities come back regrouped by frequency.
        m_of = [parallel.partition_for(list(range(mmax + 1)), r, nranks) for r in range(nranks)]
        f_of = [parallel.partition_for(list(range(nfreq)), r, nranks) for r in range(nranks)]
        skymap = None
        for mapfile in maps:
            with storage.File(mapfile, "r") as f:
                part = f["map"][local_freq[0] : local_freq[-1] + 1] if lfreq else None
            skymap = part if skymap is None else skymap + part
        alm_loc = (healpix.sphtrans_sky(skymap, lmax) if lfreq else
                   np.zeros((0, npol, lmax + 1, lmax + 1), dtype=np.complex128))       # (lfreq, npol, L, L): [l, m]
        got = parallel.exchange([np.ascontiguousarray(alm_loc[..., m_of[r]]) for r in range(nranks)])
        my_m = m_of[me]
        alm_m = np.zeros((nfreq, npol, lmax + 1, len(my_m)), dtype=np.complex128)
        for src, part in enumerate(got):
            if len(f_of[src]):
                alm_m[f_of[src]] = part
        vis_m = np.zeros((len(my_m), nfreq, 2, tel.npairs), dtype=np.complex128)
        for k, mi in enumerate(my_m):
            vis_m[k] = bt.project_vector_sky_to_telescope(mi, np.ascontiguousarray(alm_m[..., k])).reshape(nfreq, 2, tel.npairs)
        back = parallel.exchange([np.ascontiguousarray(vis_m[:, f_of[r]]) for r in range(nranks)])
        for src, part in enumerate(back):
            for k, mi in enumerate(m_of[src]):
                vis = part[k]                                  # (lfreq, 2, npairs)
                if mi == 0:
                    col_vis[..., 0] = vis[:, 0].T
                else:
                    col_vis[..., mi] = vis[:, 0].T
                    col_vis[..., -mi] = vis[:, 1].T.conj()   # conjugate only, not (-1)^m (timestream.py:759-761)

    if ndays > 0 and lfreq > 0:
        noise_ps = np.asarray(tel.noisepower(np.arange(tel.npairs)[:, np.newaxis], np.array(local_freq)[np.newaxis, :],
                                             ndays=ndays)).reshape(tel.npairs, lfreq)[:, :, np.newaxis]
        if seed is not None:
            np.random.seed(seed + parallel.rank())   # ranks must not share a noise realisation
        noise_vis = (np.array([1.0, 1.0j]) * np.random.standard_normal(col_vis.shape + (2,))).sum(axis=-1)
        noise_vis *= (noise_ps / 2.0) ** 0.5
        if seed is not None:
            np.random.seed()
        col_vis += noise_vis

    vis_stream = np.fft.ifft(col_vis, axis=-1) * ntime
    tphi = np.linspace(0, 2 * np.pi, ntime, endpoint=False)
    tstream = Timestream(outdir, m)
    for lfi, fi in enumerate(local_freq):
        os.makedirs(tstream._fdir(fi), exist_ok=True)
        with storage.File(tstream._ffile(fi), "w") as f:
            f.create_dataset("timestream", data=np.ascontiguousarray(vis_stream[:, lfi]))
            f.create_dataset("phi", data=tphi)
            f.create_dataset("feedmap", data=np.asarray(tel.feedmap))
            f.create_dataset("feedconj", data=np.asarray(tel.feedconj))
            f.create_dataset("feedmask", data=np.asarray(tel.feedmask))
            f.create_dataset("uniquepairs", data=np.asarray(tel.uniquepairs))
            f.create_dataset("baselines", data=np.asarray(tel.baselines))
            f.attrs["beamtransfer_path"] = os.path.abspath(bt.directory)
            f.attrs["ntime"] = ntime
    parallel.barrier()
    tstream.save()
    parallel.barrier()
    return tstream
