"""BeamTransfer: generation of and access to beam-transfer products, GPU-backed.

Mirrors the operator surface of the reference's ``drift.core.beamtransfer.BeamTransfer``
(drift/core/beamtransfer.py:146-1455): same constructor, config properties, file
layout, accessor and projection method names, argument meaning and return shapes.
Generation (`generate`) and the covariance projections run on the GPU through
libdriftmi; there is no CPU implementation behind them.
"""
import logging
import os
import pickle
import time

import numpy as np

from . import btgen, config, parallel, storage, util
from ._lib import block_offsets
from .device import get_context

logger = logging.getLogger(__name__)


def _contiguous_runs(ms):
    """[(first, last), ...] of a sorted list of integers."""
    runs = []
    for mi in ms:
        if runs and mi == runs[-1][1] + 1:
            runs[-1][1] = mi
        else:
            runs.append([mi, mi])
    return [(a, b) for a, b in runs]


def _find_index_sorted(a, v):
    ind = np.searchsorted(a, v)
    return ind if (ind < len(a) and v == a[ind]) else None


def _load_beam_f(path, dset_name, ind=None):
    ind = ind if ind is not None else slice(None)
    with storage.File(path, "r") as fh:
        if dset_name not in fh:
            raise RuntimeError("Malformed beam file: %s" % path)
        beam = fh[dset_name][ind]
    return np.asarray(beam)


class BeamTransfer(config.Reader):
    """Read, write and use beam transfer matrices (reference: beamtransfer.py:146).

    Parameters
    ----------
    directory : str
        Where the products live.
    telescope : TransitTelescope, optional
        If None the pickled telescope in `directory` is loaded.
    """

    mem_chunk = config.Property(proptype=float, default=3.0)
    svcut = config.Property(proptype=float, default=1e-6)
    polsvcut = config.Property(proptype=float, default=1e-4)
    # The reference's default is "bitshuffle importable" (beamtransfer.py:192); it is not in this image.  When set,
    # the blocks are truncated on the device (dm_bit_truncate_max_complex) before they are written; the files are
    # lzf-compressed either way (readable by plain h5py without the bitshuffle plugin).
    truncate = config.Property(proptype=config.truthy, default=False)
    truncate_rel = config.Property(proptype=float, default=1e-7)
    truncate_maxl = config.Property(proptype=float, default=1e-8)
    chunk_cache_size = config.Property(proptype=int, default=128)
    # MI355X-side knobs (not in the reference)
    device_chunk_gb = config.Property(proptype=float, default=6.0)  # BT-gen working set per launch group
    svd_chunk_gb = config.Property(proptype=float, default=16.0)    # SVD working set per batch of m
    beam_chunk_gb = config.Property(proptype=float, default=96.0)   # beam_m blocks resident per BT-gen call
    # weight of the ndof^3 term in the cost model of an m-block (`_m_cost`): 0.19 per generalised eigenproblem solved
    # downstream (one KLTransform; ProductManager raises it for DoubleKL and the Fisher estimators) — a least-squares fit of
    # flat + linear + cubic terms to the kernel seconds of five configs[2] shares of the round-4 build
    # (profiles/r04q_configs2_share{0,1,2,4,7}of8_generate.json: 26.5 / 28.2 / 28.4 / 30.0 / 30.7 s under the round-3
    # constants 0.6 / 0.25, which had been calibrated before the large-matrix stages got faster)
    kl_cost_weight = config.Property(proptype=float, default=0.19)
    keep_products_gb = config.Property(proptype=float, default=32.0)  # SVD products of finished batches stay in HBM up to this

    noise_weight = True
    # Opt-in stage log (bench.py's north-star share): `generate` appends (stage, [m...], wall seconds, kernel-class
    # snapshot) per BT-gen range / SVD batch / downstream (KL) batch, waiting for the device at every stage boundary.
    # None (the default): no waits beyond the ones the pipeline has anyway.
    stage_log = None

    def __init__(self, directory, telescope=None):
        self.directory = directory
        self.telescope = telescope
        self._dev = {}  # m -> dict of device tensors kept resident for the KL stage
        self._sv_host = {}  # m -> (nfreq, svd_len) singular values on the host (survive the eviction of _dev)
        if parallel.io_root() and not os.path.exists(directory):
            os.makedirs(directory)
        parallel.barrier()
        if self.telescope is None:
            try:
                with open(self._picklefile, "rb") as f:
                    self.telescope = pickle.load(f)
            except (IOError, pickle.UnpicklingError) as e:
                raise RuntimeError("Could not load Telescope object from disk.") from e

    # ---- file names (beamtransfer.py:199-224) ---------------------------------
    @property
    def _picklefile(self):
        return self.directory + "/telescopeobject.pickle"

    def _mdir(self, mi):
        return (self.directory + "/beam_m/" + util.natpattern(self.telescope.mmax)) % abs(mi)

    def _mfile(self, mi):
        return self._mdir(mi) + "/beam.hdf5"

    def _svdfile(self, mi):
        return (self.directory + "/beam_m/" + util.natpattern(self.telescope.mmax) + "/svd.hdf5") % mi

    # ---- dimensions (beamtransfer.py:1427-1453) ---------------------------------
    @property
    def ntel(self):
        return 2 * self.telescope.npairs

    @property
    def nsky(self):
        return (self.telescope.lmax + 1) * self.telescope.num_pol_sky

    @property
    def nfreq(self):
        return self.telescope.nfreq

    @property
    def svd_len(self):
        return min(self.telescope.lmax + 1, self.ntel)

    @property
    def ndofmax(self):
        return self.svd_len * self.nfreq

    def ndof(self, mi):
        return self._svd_num(mi)[1][-1]

    # ---- accessors ------------------------------------------------------------------
    @util.cache_last
    def beam_m(self, mi, fi=None):
        """(nfreq, 2, npairs, npol_sky, lmax+1) beam transfer block of one m
        (or one frequency of it), zero where skipped (beamtransfer.py:257-308)."""
        tel = self.telescope
        ind_list = [np.arange(2), tel.included_baseline, tel.included_pol, np.arange(mi, tel.lmax + 1)]
        shape = (2, tel.nbase, tel.num_pol_sky, tel.lmax + 1)
        if fi is None:
            ind_list = [tel.included_freq] + ind_list
            shape = (tel.nfreq,) + shape
        bf = np.zeros(shape, dtype=np.complex128)
        if fi is not None:
            fi = _find_index_sorted(tel.included_freq, fi)
            if fi is None:
                return bf
        data = _load_beam_f(self._mfile(mi), "beam_m", fi)
        full = (len(tel.included_baseline) == tel.nbase and len(tel.included_pol) == tel.num_pol_sky
                and (fi is not None or len(tel.included_freq) == tel.nfreq))
        if full:   # nothing skipped: a plain slice assignment (the general scatter walks the block element by element)
            bf[..., mi:] = data
        else:
            bf[np.ix_(*ind_list)] = data
        return bf

    @util.cache_last
    def beam_svd(self, mi, fi=None):
        """(nfreq, svd_len, npol_sky, lmax+1) SVD beam (sky -> SVD basis), full spectrum."""
        return _load_beam_f(self._svdfile(mi), "beam_svd", fi)

    @util.cache_last
    def invbeam_svd(self, mi, fi=None):
        """(nfreq, npol_sky, lmax+1, svd_len) pseudo-inverse of the SVD beam."""
        return _load_beam_f(self._svdfile(mi), "invbeam_svd", fi)

    @util.cache_last
    def beam_ut(self, mi, fi=None):
        """(nfreq, svd_len, ntel) telescope -> SVD basis."""
        return _load_beam_f(self._svdfile(mi), "beam_ut", fi)

    @util.cache_last
    def beam_singularvalues(self, mi):
        """(nfreq, svd_len) singular values."""
        return _load_beam_f(self._svdfile(mi), "singularvalues")

    def svd_all(self):
        with storage.File(self.directory + "/svdspectrum.hdf5", "r") as f:
            return f["singularvalues"][:]

    # ---- generation -------------------------------------------------------------------
    def _stage_begin(self):
        if self.stage_log is None:
            return None
        ctx = get_context()
        ctx.sync()
        return (time.perf_counter(), ctx.prof_report() if ctx.prof_enabled() else {})

    def _stage_end(self, tok, stage, ms):
        """One record of the stage log: wall seconds between two device-idle points and the kernel-class times /
        work counters that accrued in between (dm_prof_report differences)."""
        if tok is None:
            return
        ctx = get_context()
        ctx.sync()
        dt = time.perf_counter() - tok[0]
        now = ctx.prof_report() if ctx.prof_enabled() else {}
        cls = {}
        for k, v in now.items():
            o = tok[1].get(k, dict(ms=0.0, flops=0.0, launches=0))
            d = dict(ms=v["ms"] - o["ms"], flops=v["flops"] - o["flops"], launches=v["launches"] - o["launches"])
            if d["launches"] > 0:
                cls[k] = d
        self.stage_log.append(dict(stage=stage, ms=[int(m) for m in ms], seconds=dt, classes=cls))

    def generate(self, regen=False, skip_svd=False, skip_svd_inv=False, after_batch=None):
        """Generate and save all products (beamtransfer.py:447-480).

        The reference runs the stages one after the other over all m, through the files.  Here a rank owns ONE
        contiguous, cost-balanced range of m (`_my_ms`) and takes it through the stages while the blocks are resident in
        HBM: beam-transfer generation of as many of its m as `beam_chunk_gb` holds, then batch by batch (`svd_chunk_gb`)
        the SVD chain on the resident blocks and — through `after_batch(ms)`, which `ProductManager.generate` uses for the
        KL transforms — everything downstream that wants the SVD products of those m.  Finished products go to the
        writer pool; nothing is read back from a file during generation."""
        st = time.time()
        self._generate_dirs()
        if parallel.io_root() and not storage.discard():
            with open(self._picklefile, "wb") as f:
                pickle.dump(self.telescope, f)
        tel = self.telescope
        ctx = get_context()
        marker = self.directory + "/beam_m/COMPLETED"
        mine = self._my_ms()
        per_block = tel.nfreq * 2 * tel.nbase * tel.num_pol_sky * (tel.lmax + 1) * 16
        nb_max = max(1, int(self.beam_chunk_gb * (1 << 30) // per_block))
        runs = _contiguous_runs(mine)
        ranges = [(a0, min(a0 + nb_max, b + 1) - 1) for (a, b) in runs for a0 in range(a, b + 1, nb_max)]
        have_beams = os.path.exists(marker) and not regen
        self._beam_all = None
        for (a, b) in ranges:
            ms = list(range(a, b + 1))
            need_bt = not have_beams and (regen or not all(os.path.exists(self._mfile(mi)) for mi in ms))
            need_svd = not skip_svd and (regen or not all(storage.can_open(self._svdfile(mi)) for mi in ms))
            beam_all = None
            if need_bt:
                whole = a == 0 and b == tel.mmax
                # the writer queue may still hold views of the previous range's blocks (they are `resident`: outside its
                # bound on device bytes) — let their host copies finish before a second chunk of up to beam_chunk_gb is allocated
                storage.wait_copies()
                tok = self._stage_begin()
                beam_all = btgen.beam_m_all(tel, ctx=ctx, max_bytes=int(self.device_chunk_gb * (1 << 30)),
                                            m_range=None if whole else (a, b))
                if self.truncate:
                    # beamtransfer.py:641-646: rows of the m-ordered array (runs over l) truncated to
                    # max(truncate_rel |z|, truncate_maxl max_l |z|); in place, so the SVD stage sees what the files hold
                    ctx.bit_truncate_max_complex(beam_all, self.truncate_rel, self.truncate_maxl)
                self._stage_end(tok, "btgen", ms)
                self._write_beam_files(beam_all, a, b, regen)
            self._beam_all, self._beam_all_m0 = beam_all, a
            if need_svd:
                self._svd_batches(ms, regen, skip_svd_inv, after_batch)
            elif after_batch is not None and not skip_svd:
                for batch in self._svd_batch_lists(ms):
                    tok = self._stage_begin()
                    after_batch(batch)
                    self._stage_end(tok, "kl", batch)
                    self._evict(batch)
            self._beam_all = None
            del beam_all
        storage.flush()
        parallel.barrier()
        if parallel.rank0() and not storage.discard() and not parallel.is_virtual():
            open(marker, "a").close()     # (an emulated share is not the whole job: no marker)
        if not skip_svd:
            self._collect_svd_spectrum()
        parallel.barrier()
        if parallel.rank0():
            logger.info("Beam generation time: %f" % (time.time() - st))

    generate_cache = generate  # beamtransfer.py: old name kept by the reference

    def _generate_dirs(self):
        if parallel.io_root():
            os.makedirs(self.directory, exist_ok=True)
            for mi in range(self.telescope.mmax + 1):
                os.makedirs(self._mdir(mi), exist_ok=True)
        parallel.barrier()

    def _m_cost(self, m):
        """Relative cost of one m-block through the whole path: beam-transfer generation is about the same for every m,
        the SVD chain follows the number of l >= m, the KL stage its cube (ndof falls with m)."""
        x = float(self.telescope.lmax + 1 - m) / float(self.telescope.lmax + 1)
        # Round 5: the SVD chain works on the columns l >= m only and forms cross Gram blocks, so its cost falls faster
        # with m than the linear term of rounds 3-4 said.  The coefficients are what reproduces the boundaries that the
        # MEASURED seconds per block of full sets of configs[2] shares ask for (the per-share densities re-partitioned;
        # grid search over the coefficients with the SHT coupling of `_my_ms` in the loop): first on
        # profiles/r05t_configs2_shares.json and the set before it, then — SVD1 / SVD2 as subspace phases took a quarter off
        # the wide chains and little off the tall ones — on the set of that build, which asks for (0, 25) (26, 54) (55, 86)
        # (87, 123) (124, 170) (171, 231) (232, 316) (317, 512).  0.19 units x^3 are the KLTransform's (`kl_cost_weight`:
        # DoubleKL and the Fisher stage raise it).
        tel = self.telescope
        P, T = int(tel.num_pol_sky), int(self.ntel)
        if P > 1 and P * (tel.lmax + 1 - m) * 100 <= T * 95:
            # a TALL block (more rows than sky columns l >= m: dm_svd_chain_lmin takes SVD1 through the transposed matrix, a
            # P (L - m)-square Gram eigenproblem instead of a T-square one): close to linear in x — SVD batches at
            # m = 300 / 400 / 480 take 0.19 / 0.10 / 0.015 s per block (scratch/svd_phase_probe.py), BT-gen 0.024 s, less
            # what the ring transform skips at high m (bt_ring_skip_lookup: 1.5 s of the last share)
            return 0.92 * x + self.kl_cost_weight * x ** 3
        return 0.10 + 0.60 * x + (0.36 + self.kl_cost_weight) * x ** 3

    def _my_ms(self, mlist=None):
        """m-blocks owned by this rank: ONE contiguous range with (nearly) the same summed cost on every rank
        (`parallel.partition_contiguous`) — a rank keeps its blocks in HBM from generation to the KL stage, and
        beam-transfer generation produces a contiguous range of m per call.  (The reference's split is contiguous with
        equal counts, beamtransfer.py:720-722; any assignment gives the same files.)"""
        tel = self.telescope
        mlist = list(range(tel.mmax + 1)) if mlist is None else list(mlist)
        costs = np.array([self._m_cost(m) for m in mlist], dtype=np.float64)
        n = parallel.size()
        mcut = self._sht_mcut() if n > 1 and mlist == list(range(mlist[0], mlist[-1] + 1)) else -1
        if mcut >= 0:
            # The SHT refinement couples the m <= mcut through the polar rings: a rank whose range starts among them
            # transforms ALL of them.  That is a cost per RANK, not per m — a few fixed-point passes
            # spread it over the rank's own blocks and move the boundaries (every rank computes the same partition).  A block
            # transformed on the side costs about 0.39 of the flat BT-gen term of `_m_cost` (calibrated on the configs[2]
            # shares, profiles/r04b_configs2_share*of8_generate.json: 0.035 s per block; 0.15 units are 0.09 s since the
            # round-5 refit of `_m_cost`).
            base = costs.copy()
            for _ in range(4):
                extra = np.zeros_like(base)
                for r in range(n):
                    mine = parallel.partition_contiguous(mlist, costs, n=n, r=r)
                    if not mine or mine[0] > mcut:
                        continue
                    side = (min(mcut, mlist[-1]) + 1 - mlist[0]) - sum(1 for m in mine if m <= mcut)
                    i0 = mine[0] - mlist[0]
                    extra[i0 : i0 + len(mine)] = 0.39 * 0.15 * max(side, 0) / len(mine)
                costs = base + extra
        return parallel.partition_contiguous(mlist, list(costs))

    def _sht_mcut(self):
        """Largest m the SHT refinement couples to other m (over the nside groups of the telescope's columns); -1 without
        refinement.  Host arithmetic only (`dm_bt_alias_info`), remembered."""
        tel = self.telescope
        if not int(getattr(tel, "sht_iter", 0) or 0):
            return -1
        memo = self.__dict__.setdefault("_mcut_memo", {})
        key = (int(tel.sht_iter), int(tel.lmax), int(tel.nfreq), int(tel.nbase))
        if key not in memo:
            from . import healpix
            from ._lib import bt_alias_info

            ff, bb = np.meshgrid(tel.included_freq, tel.included_baseline, indexing="ij")
            lmax_bf, _ = tel.baseline_lmax(bb.ravel(), ff.ravel())
            nsides = btgen._nside_of(tel, lmax_bf)
            mc = -1
            for ns in np.unique(nsides):
                cth, sth = healpix.ring_trig(int(ns))
                mc = max(mc, bt_alias_info(int(ns), cth, sth, tel.num_pol_sky > 1, int(lmax_bf[nsides == ns].max()))[1])
            memo[key] = int(mc)
        return memo[key]

    def _write_beam_files(self, beam_all, a, b, regen):
        """Hand the beam_m files of m = a..b (device tensor, first axis = m - a) to the writer pool."""
        if storage.discard():
            return
        tel = self.telescope
        ctx = get_context()
        finc, binc, pinc = tel.included_freq, tel.included_baseline, tel.included_pol

        def write_m(mi, blk):
            with storage.File(self._mfile(mi), "w") as f:
                # the included frequencies / baselines / polarisations and l >= m (beamtransfer.py:567-577, :649-663): plain
                # slices where nothing is skipped — a 5-axis fancy index copies a configs[2] block (1.8 GB) element by
                # element under the GIL, six seconds per file, and was what bounded the file output of a rank
                data = blk[..., mi:]
                for ax, inc, full in ((0, finc, tel.nfreq), (2, binc, tel.nbase), (3, pinc, tel.num_pol_sky)):
                    inc = np.asarray(inc)
                    if inc.size != full or not np.array_equal(inc, np.arange(full)):
                        data = np.take(data, inc, axis=ax)
                # chunk shape and compression of beamtransfer.py:548-571
                f.create_dataset("beam_m", data=data, **storage.compression_kwargs(
                    (1, 2, min(10, len(binc)), len(pinc), tel.lmax + 1 - mi)))
                f.attrs["m"] = mi
                f.attrs["frequencies"] = tel.frequencies
                # not in the reference's files: which reading of its SHT (healpy.map2alm through cora, neither readable
                # here) these blocks were made with — DESIGN.md section 3
                f.attrs["sht_iter"] = int(getattr(tel, "sht_iter", 0) or 0)
                f.attrs["sht_ring_weights"] = bool(getattr(tel, "sht_ring_weights", None) is not None)
                # calls covering at most this many m take the belt rings through the matrix form instead of the FFT
                # (DM_BT_NARROW): blocks made by such calls equal those of wide calls to rounding (2e-13), not bit for bit
                # — with truncation on, the quantised files can therefore depend on beam_chunk_gb / the rank count
                f.attrs["sht_narrow_max_m"] = int(os.environ.get("DM_BT_NARROW", "8"))

        # the host copies are made by the writer pool's copy thread (storage.Deferred) while the SVD stage reads the same
        # blocks: nothing writes to `beam_all` after this point
        ev = ctx.record_event()
        for mi in range(a, b + 1):
            if os.path.exists(self._mfile(mi)) and not regen:
                continue
            storage.submit(write_m, mi, ctx.defer_host(beam_all[mi - a], ev, resident=True))

    def _generate_mfiles(self, regen=False):
        """beam_m files alone (beamtransfer.py:502-676), for callers that want the stage by itself; `generate` runs
        it fused with the SVD stage.  The reference computes (f, b) chunks on each rank and transposes to m-order with
        an all-to-all; here every rank synthesises the maps it needs and keeps only its own m-blocks — map synthesis is
        a few percent of the per-m cost, so replicating it is cheaper than an exchange step."""
        marker = self.directory + "/beam_m/COMPLETED"
        if os.path.exists(marker) and not regen:
            return
        tel = self.telescope
        ctx = get_context()
        st = time.time()
        mine = self._my_ms()
        per_block = tel.nfreq * 2 * tel.nbase * tel.num_pol_sky * (tel.lmax + 1) * 16
        nb_max = max(1, int(self.beam_chunk_gb * (1 << 30) // per_block))
        runs = _contiguous_runs(mine)
        ranges = [(a0, min(a0 + nb_max, b + 1) - 1) for (a, b) in runs for a0 in range(a, b + 1, nb_max)]
        self._beam_all = None
        for (a, b) in ranges:
            whole = a == 0 and b == tel.mmax
            beam_all = btgen.beam_m_all(tel, ctx=ctx, max_bytes=int(self.device_chunk_gb * (1 << 30)),
                                        m_range=None if whole else (a, b))
            if self.truncate:
                ctx.bit_truncate_max_complex(beam_all, self.truncate_rel, self.truncate_maxl)
            if whole:
                self._beam_all, self._beam_all_m0 = beam_all, 0  # kept for a following _generate_svdfiles
            self._write_beam_files(beam_all, a, b, regen)
            del beam_all
        storage.flush()
        parallel.barrier()
        if parallel.rank0():
            if not parallel.is_virtual():
                open(marker, "a").close()
            logger.info("=== beam_m generation took %f s ===" % (time.time() - st))

    def _noisew(self):
        """(F, T) noise weights noisepower^-1/2, duplicated for the two m signs (beamtransfer.py:810-813)."""
        tel = self.telescope
        nw = np.array([np.asarray(tel.noisepower(np.arange(tel.npairs), fi)).reshape(-1) ** -0.5
                       for fi in range(tel.nfreq)])
        return np.concatenate([nw, nw], axis=1)

    def _noisew_device(self):
        """The noise weights on the device, uploaded once per context: a pageable host-to-device copy waits for the
        stream to drain, which would serialise every batch behind the previous one's kernels."""
        ctx = get_context()
        key = id(ctx)
        cache = self.__dict__.setdefault("_noisew_dev", {})
        if key not in cache:
            if len(cache) > 8:
                cache.clear()
            cache[key] = ctx.to_device(self._noisew())
        return cache[key]

    def svd_device(self, beam_blocks, skip_svd_inv=False, ms=None):
        """Run the SVD chain on a device tensor (nblk, F, 2, B, P, L); returns the dict of
        device products (see Context.svd_chain).  ``ms``: the m of each block — an m-block is zero for l < m
        (`beam_m`, beamtransfer.py:257-308), and the chains then skip those columns."""
        ctx = get_context()
        nblk, F = int(beam_blocks.shape[0]), int(beam_blocks.shape[1])
        T, P, L = self.ntel, self.telescope.num_pol_sky, self.telescope.lmax + 1
        nw = self._noisew_device()
        return ctx.svd_chain(beam_blocks.reshape(nblk, F, T, P, L), nw, self.polsvcut, skip_svd_inv=skip_svd_inv,
                             lmin=None if ms is None else [int(m) for m in ms])

    def _svd_batch_lists(self, ms):
        tel = self.telescope
        F, T, P, L, K = tel.nfreq, self.ntel, tel.num_pol_sky, tel.lmax + 1, self.svd_len
        per_m = F * (T * (P * L + T) * 2 + K * P * L * 2 + K * T) * 16
        nb = max(1, int(self.svd_chunk_gb * (1 << 30) // per_m))
        ms = list(ms)
        # as many batches as the budget asks for, of equal size: a short last batch runs the same lock-step launch chains
        # for a fraction of the work (and falls under the batch sizes where the two-stage tridiagonalisation pays)
        nbat = max(1, -(-len(ms) // nb))
        base, rem = divmod(len(ms), nbat)   # sizes differ by at most one: 34 blocks in three batches are 12 + 11 + 11
        out, c0 = [], 0
        for k in range(nbat):
            n = base + (1 if k < rem else 0)
            if n:
                out.append(ms[c0 : c0 + n])
            c0 += n
        return out

    def _svd_batches(self, ms, regen=False, skip_svd_inv=False, after_batch=None):
        """SVD chain of the given m in batches that fit `svd_chunk_gb` (beamtransfer.py:730-929): the products stay on
        the device (`_dev`) for whoever comes next, the files go to the writer pool; `after_batch(ms)` runs right after a
        batch while its products are resident, then they are evicted (unless `keep_products`)."""
        tel = self.telescope
        ctx = get_context()
        T, P, L, K = self.ntel, tel.num_pol_sky, tel.lmax + 1, self.svd_len
        todo = [mi for mi in ms if regen or not storage.can_open(self._svdfile(mi))]
        for batch in self._svd_batch_lists(todo):
            tok = self._stage_begin()
            blocks = self._device_beam_blocks(batch)
            res = self.svd_device(blocks, skip_svd_inv=skip_svd_inv, ms=batch)
            del blocks
            sv_host = ctx.to_host(res["singularvalues"])
            self._stage_end(tok, "svd", batch)
            if not storage.discard():
                ev = ctx.record_event()   # host copies by the copy thread (storage.Deferred); the products are read-only from here

                def write_svd(mi, bsvd, ibsvd, but, sig):
                    with storage.File(self._svdfile(mi), "w") as fs:   # chunk shapes of beamtransfer.py:741-798
                        k10 = min(10, K)
                        fs.create_dataset("beam_svd", data=bsvd, **storage.compression_kwargs((1, k10, P, L)))
                        if ibsvd is not None:
                            fs.create_dataset("invbeam_svd", data=ibsvd, **storage.compression_kwargs((1, P, L, k10)))
                        fs.create_dataset("beam_ut", data=but, **storage.compression_kwargs((1, k10, T)))
                        fs.create_dataset("singularvalues", data=sig)
                        fs.attrs["baselines"] = tel.baselines
                        fs.attrs["m"] = mi
                        fs.attrs["frequencies"] = tel.frequencies

            views = self._register_sv(batch, sv_host)
            bsvd_v, but_v = res["beam_svd"].unbind(0), res["beam_ut"].unbind(0)   # all per-m views in one call each
            for i, mi in enumerate(batch):
                self._sv_host[mi] = views[i]
                self._dev[mi] = dict(beam_svd=bsvd_v[i], beam_ut=but_v[i], singularvalues=views[i])
                if not storage.discard():
                    storage.submit(write_svd, mi, ctx.defer_host(bsvd_v[i], ev),
                                   None if skip_svd_inv else ctx.defer_host(res["invbeam_svd"][i], ev),
                                   ctx.defer_host(but_v[i], ev), sv_host[i])
            del res
            if after_batch is not None:
                tok = self._stage_begin()
                after_batch(batch)
                self._stage_end(tok, "kl", batch)
                self._evict(batch)

    def _evict(self, ms):
        """Drop the device copies of the SVD products of these m (they are re-read from their files when wanted again);
        the singular values stay on the host for `_svd_num` and the spectrum file."""
        tel = self.telescope
        per_m = tel.nfreq * self.svd_len * (tel.num_pol_sky * (tel.lmax + 1) + self.ntel) * 16   # beam_svd + beam_ut
        if per_m * len(self._dev) <= self.keep_products_gb * (1 << 30):
            return
        for mi in ms:
            self._dev.pop(mi, None)
        self.__dict__.pop("_stack_memo", None)   # its views would keep the evicted batch alive

    def _generate_svdfiles(self, regen=False, skip_svd_inv=False):
        """svd.hdf5 for every m of this rank (beamtransfer.py:678-728, :730-929) as a stage of its own."""
        self._svd_batches(self._my_ms(), regen, skip_svd_inv)
        storage.flush()
        parallel.barrier()
        self._collect_svd_spectrum()

    def _device_beam_blocks(self, ms):
        """(len(ms), F, 2, B, P, L) device tensor of the given m-blocks: from the resident
        generation result when it covers them, else re-read from the beam_m files."""
        ctx = get_context()
        ba = getattr(self, "_beam_all", None)
        if ba is not None:
            m0 = getattr(self, "_beam_all_m0", 0)
            if all(m0 <= mi < m0 + int(ba.shape[0]) for mi in ms):
                ms = list(ms)
                if ms == list(range(ms[0], ms[0] + len(ms))):
                    return ba[ms[0] - m0 : ms[0] - m0 + len(ms)]   # a view: no copy of the blocks
                import torch

                return ba[torch.as_tensor([mi - m0 for mi in ms], device=ba.device)]
        return ctx.to_device(np.stack([self.beam_m(mi) for mi in ms]))

    def _collect_svd_spectrum(self):
        """svdspectrum.hdf5: (mmax+1, nfreq, svd_len) (beamtransfer.py:931-947)."""
        mine = [(mi, self._sv_host[mi] if mi in self._sv_host else self.beam_singularvalues(mi)) for mi in self._my_ms()]
        allparts = parallel.gather_objects(mine)
        if parallel.rank0():
            spec = np.zeros((self.telescope.mmax + 1, self.nfreq, self.svd_len))
            for part in allparts:
                for mi, sv in part:
                    spec[mi] = sv
            if not storage.discard() and not parallel.is_virtual():
                with storage.File(self.directory + "/svdspectrum.hdf5", "w") as f:
                    f.create_dataset("singularvalues", data=spec)
        parallel.barrier()

    # ---- SVD bookkeeping (beamtransfer.py:1116-1133) ------------------------------------
    def _svd_num(self, mi):
        """(svnum, svbounds) of one m (beamtransfer.py:1116-1133); memoised per singular-value array,
        the KL stage asks for it several times per m."""
        sv = self._sv_host[mi] if mi in self._sv_host else (
            self._dev[mi]["singularvalues"] if mi in self._dev else self.beam_singularvalues(mi))
        memo = self.__dict__.setdefault("_svnum_memo", {})
        hit = memo.get(mi)
        if hit is not None and hit[0] is sv and hit[1] == self.svcut:
            return hit[2], hit[3]
        svnum = (sv > sv.max() * self.svcut).sum(axis=1)
        svbounds = np.zeros(svnum.size + 1, dtype=svnum.dtype)
        np.cumsum(svnum, out=svbounds[1:])
        memo[mi] = (sv, self.svcut, svnum, svbounds)
        return svnum, svbounds

    def _register_sv(self, ms, sv_all):
        """Per-m views of the (len(ms), nfreq, svd_len) singular values of a batch, with the (svnum, svbounds) of every m
        computed in one vectorised pass — the KL stage asks for them several times per m, and 10^2..10^3 tiny numpy
        calls between two stages are a millisecond of idle GPU."""
        sv_all = np.asarray(sv_all)
        memo = self.__dict__.setdefault("_svnum_memo", {})
        top = sv_all.reshape(sv_all.shape[0], -1).max(axis=1) if sv_all.size else np.zeros(sv_all.shape[0])
        svnum = (sv_all > (top * self.svcut)[:, None, None]).sum(axis=2)
        bounds = np.zeros((sv_all.shape[0], sv_all.shape[1] + 1), dtype=svnum.dtype)
        np.cumsum(svnum, axis=1, out=bounds[:, 1:])
        views = []
        for i, mi in enumerate(ms):
            v = sv_all[i]
            memo[mi] = (v, self.svcut, svnum[i], bounds[i])
            views.append(v)
        return views

    def _svd_freq_iter(self, mi):
        num = self._svd_num(mi)[0]
        return [fi for fi in range(self.nfreq) if num[fi] > 0]

    # ---- covariance projections (GPU) ----------------------------------------------------
    def _dev_products(self, mi):
        ctx = get_context()
        if mi not in self._dev:
            self._dev[mi] = dict(beam_svd=ctx.to_device(self.beam_svd(mi)), beam_ut=ctx.to_device(self.beam_ut(mi)),
                                 singularvalues=self.beam_singularvalues(mi))
        return self._dev[mi]

    @staticmethod
    def _cl_device(mat):
        """(P,P,L,F,F) real sky covariance -> device (P,P,F,F,L), mask of the non-zero pol pairs, and whether
        the array is symmetric under f <-> f' (only then may dm_project_cov mirror the frequency blocks; the
        reference, beamtransfer.py:1135-1188, takes any array)."""
        ctx = get_context()
        key = id(mat)
        cache = BeamTransfer._clcache
        if key not in cache or cache[key][0] is not mat:
            # uploaded as it lies; the pol-pair mask, the symmetry test and the (.., F, F, L) layout are made on the device
            # (on the host the strided transposition of a configs[2] table — 270 MB — takes two seconds, with the GPU idle
            # in front of the first KL batch)
            import torch

            P = mat.shape[0]
            dev0 = ctx.to_device(np.ascontiguousarray(np.asarray(mat, dtype=np.float64)))
            mask = (dev0.abs().reshape(P, P, -1).amax(dim=-1) > 0).to(torch.int32).cpu().numpy()
            sym = bool(torch.equal(dev0, dev0.transpose(3, 4)))
            dev = dev0.permute(0, 1, 3, 4, 2).contiguous()
            del dev0
            if len(cache) > 8:
                cache.clear()
            cache[key] = (mat, dev, mask, sym)
        return cache[key][1], cache[key][2], cache[key][3]

    _clcache = {}

    def _stacked_products(self, ms, key):
        """The per-m device products `key` of a batch as ONE tensor (len(ms), ...).  The products of a batch are views of
        the SVD chain's output, back to back in memory: then the stack is a view as well (no copy, no per-m Python work);
        the last result is remembered, the KL stage asks for `beam_svd` once per covariance."""
        import torch

        memo = self.__dict__.setdefault("_stack_memo", {})
        tens = [self._dev_products(mi)[key] for mi in ms]
        if not tens:
            return torch.empty((0,), dtype=torch.complex128, device="cuda")
        sig = (key, tuple(ms), tuple(t.data_ptr() for t in tens[:2]), tens[-1].data_ptr())
        hit = memo.get(key)
        if hit is not None and hit[0] == sig:
            return hit[1]
        t0 = tens[0]
        base, o0, ne, shp = t0._base, t0.storage_offset(), t0.numel(), t0.shape
        # views of one base tensor, equally spaced and dense (what `unbind` of a batch result gives)
        same = base is not None and t0.is_contiguous() and all(
            t._base is base and t.shape == shp and t.storage_offset() == o0 + i * ne and t.stride() == t0.stride()
            for i, t in enumerate(tens))
        if same and t0.numel() > 0:
            out = torch.as_strided(t0, (len(tens),) + tuple(t0.shape), (t0.numel(),) + tuple(t0.stride()))
            memo[key] = (sig, out)   # a VIEW of the batch result: valid as long as the addresses in `sig` are
            return out
        memo.pop(key, None)          # a stacked COPY is never remembered (the allocator may hand the same addresses out again)
        return torch.stack(tens)

    def project_matrix_sky_to_svd_device(self, ms, mat, out, off, temponly=False, zero_first=True):
        """Batched form: project `mat` for all m in `ms` into the flat device buffer `out`."""
        ctx = get_context()
        bsvd = self._stacked_products(ms, "beam_svd")
        svnum = np.stack([self._svd_num(mi)[0] for mi in ms])
        cl, mask, sym = self._cl_device(mat)
        ctx.project_cov(bsvd, svnum, cl, out, off, npol=1 if temponly else None, polmask=mask, l0=np.array(ms),
                        zero_first=zero_first, symmetric=sym)

    def project_matrix_sky_to_svd(self, mi, mat, temponly=False):
        """Sky covariance [pol, pol, l, freq, freq] -> SVD basis [nsvd, nsvd]
        (beamtransfer.py:1135-1188)."""
        ctx = get_context()
        n = int(self.ndof(mi))
        off, tot = block_offsets([n])
        out = ctx.empty((max(tot, 1),), np.complex128)
        self.project_matrix_sky_to_svd_device([mi], mat, out, off, temponly=temponly)
        ctx.sync()
        return out[: n * n].cpu().numpy().reshape(n, n)

    def project_matrix_diagonal_telescope_to_svd(self, mi, dmat):
        """Diagonal telescope-basis matrix [nfreq, ntel] -> SVD basis (beamtransfer.py:1190-1231)."""
        ctx = get_context()
        n = int(self.ndof(mi))
        off, tot = block_offsets([n])
        out = ctx.empty((max(tot, 1),), np.complex128)
        p = self._dev_products(mi)
        ctx.project_diag(p["beam_ut"][None], self._svd_num(mi)[0][None], ctx.to_device(np.asarray(dmat, dtype=np.float64)),
                         out, off, alpha=1.0, accumulate=False)
        ctx.sync()
        return out[: n * n].cpu().numpy().reshape(n, n)

    # ---- pseudo-inverse of the full beam, map-making operators -------------------------------
    noise_weight = True  # beamtransfer.py:314

    def invbeam_m(self, mi):
        """Moore-Penrose pseudo-inverse of the beam of one m, (nfreq, npol_sky, lmax+1, ntel)
        (beamtransfer.py:317-358: pinv with rcond 1e-6 of the blocks weighted by the noise of
        frequency 0).  On the device: one-sided Jacobi on [w B | I] gives W (w B) = S V^H with the
        accumulated W riding along, so pinv = (S V^H)^H S^-2 W, one grouped ZGEMM with the cut
        sigma > 1e-6 sigma_max folded into the contraction weights."""
        tel = self.telescope
        ctx = get_context()
        F, T, S = self.nfreq, self.ntel, self.nsky
        beam = self.beam_m(mi).reshape(F, 2, tel.npairs, S)
        noisew = None
        if self.noise_weight:
            noisew = np.asarray(tel.noisepower(np.arange(tel.npairs), 0)).flatten() ** (-0.5)
            beam = beam * noisew[np.newaxis, np.newaxis, :, np.newaxis]
        beam = beam.reshape(F, T, S)
        Z = np.zeros((F, T, S + T), dtype=np.complex128)
        Z[:, :, :S] = beam
        Z[:, :, S:] = np.eye(T)
        dZ = ctx.to_device(Z)
        sigma, _ = ctx.jacobi_rows(dZ, T, S + T, 0, S, S + T, stride=T * (S + T), batch=F)
        ctx.sync()
        sg = sigma.cpu().numpy()[:, :T]
        smax = sg.max(axis=1, keepdims=True)
        wts = np.where((sg > 1e-6 * smax) & (sg > 0.0), 1.0 / np.where(sg > 0.0, sg, 1.0) ** 2, 0.0)
        dW = ctx.to_device(np.ascontiguousarray(wts))
        out = ctx.empty((F, S, T), np.complex128)
        # out[f] = (S V^H)^H diag(w) W :  A(m = sky, k = row) = conj(Z[f, k, m]),  B(k, n) = Z[f, k, S + n]
        ctx.zgemm(dZ, dZ[:, :, S:], out, S, T, T, rsA=1, csA=S + T, rsB=S + T, csB=1, ldc=T, conjA=True, kscale=dW,
                  batch=F, strideA=T * (S + T), strideB=T * (S + T), strideC=S * T, stride_kscale=T)
        ctx.sync()
        ib = out.cpu().numpy()
        if self.noise_weight:
            ib = (ib.reshape(-1, tel.npairs) * noisew).reshape(F, S, T)
        return ib.reshape(F, tel.num_pol_sky, tel.lmax + 1, T)

    def project_vector_telescope_to_sky(self, mi, vec):
        """Map-making: [nfreq, ntel] -> [nfreq, npol, lmax+1] (beamtransfer.py:1014-1046)."""
        tel = self.telescope
        vec = np.asarray(vec).reshape(self.nfreq, self.ntel)
        if np.all(vec == 0):
            return np.zeros((self.nfreq, tel.num_pol_sky, tel.lmax + 1), dtype=np.complex128)
        ib = self.invbeam_m(mi).reshape(self.nfreq, self.nsky, self.ntel)
        return _device_bgemv(ib, vec).reshape(self.nfreq, tel.num_pol_sky, tel.lmax + 1)

    project_vector_backward = project_vector_telescope_to_sky

    def project_vector_backward_dirty(self, mi, vec):
        """Dirty back-projection B^H (v / diag(B B^H)) (beamtransfer.py:1050-1072)."""
        tel = self.telescope
        vec = np.asarray(vec).reshape(self.nfreq, self.ntel)
        if np.all(vec == 0):
            return np.zeros((self.nfreq, tel.num_pol_sky, tel.lmax + 1), dtype=np.complex128)
        beam = self.beam_m(mi).reshape(self.nfreq, self.ntel, self.nsky)
        norm = np.einsum("ftk,ftk->ft", beam, beam.conj())
        norm = np.where(norm < 1e-6, 0.0, 1.0 / np.where(norm == 0.0, 1.0, norm))
        dbeam = np.ascontiguousarray(beam.transpose(0, 2, 1).conj())
        return _device_bgemv(dbeam, vec * norm).reshape(self.nfreq, tel.num_pol_sky, tel.lmax + 1)

    def project_matrix_sky_to_telescope(self, mi, mat, temponly=False):
        """Sky covariance [pol, pol, l, f, f'] -> visibility basis [nfreq, ntel, nfreq, ntel]
        (beamtransfer.py:1074-1112): the same grouped projection as the SVD-basis one with the
        un-compressed beam (every frequency keeps all ntel rows)."""
        tel = self.telescope
        ctx = get_context()
        F, T, P, L = self.nfreq, self.ntel, tel.num_pol_sky, tel.lmax + 1
        n = F * T
        off, tot = block_offsets([n])
        out = ctx.empty((max(tot, 1),), np.complex128)
        beam = ctx.to_device(np.ascontiguousarray(self.beam_m(mi).reshape(1, F, T, P, L)))
        svnum = np.full((1, F), T, dtype=np.int32)
        cl, mask, sym = self._cl_device(mat)
        ctx.project_cov(beam, svnum, cl, out, off, npol=1 if temponly else None, polmask=mask, l0=np.array([mi]),
                        zero_first=True, symmetric=sym)
        ctx.sync()
        return out[: n * n].cpu().numpy().reshape(F, T, F, T)

    # ---- vector projections (light, host-side views of stored blocks) -----------------------
    def project_vector_sky_to_telescope(self, mi, vec):
        """[nfreq, npol, lmax+1] -> [nfreq, ntel] (beamtransfer.py:970-1010)."""
        tel = self.telescope
        vecf = np.zeros((self.nfreq, 2, tel.nbase), dtype=np.complex128)
        beam = self.beam_m(mi).reshape(self.nfreq, self.ntel, self.nsky)
        v = np.asarray(vec).reshape(self.nfreq, self.nsky)
        if np.all(v == 0):
            return vecf.reshape(self.nfreq, self.ntel)
        return _device_bgemv(beam, v).reshape(self.nfreq, self.ntel)

    project_vector_forward = project_vector_sky_to_telescope

    def project_vector_telescope_to_svd(self, mi, vec):
        """[nfreq, ntel, ...] -> [nsvd, ...] (beamtransfer.py:1233-1271)."""
        svnum, svbounds = self._svd_num(mi)
        vec = np.asarray(vec)
        vecf = np.zeros((svbounds[-1],) + vec.shape[2:], dtype=np.complex128)
        if np.all(vec == 0):
            return vecf
        beam = self.beam_ut(mi)
        for fi in self._svd_freq_iter(mi):
            vecf[svbounds[fi] : svbounds[fi + 1]] = _device_gemm(beam[fi, : svnum[fi], :], vec[fi])
        return vecf

    def project_vector_svd_to_telescope(self, mi, svec):
        """[nsvd] -> [nfreq, 2, npairs] (beamtransfer.py:1273-1322)."""
        tel = self.telescope
        svnum, svbounds = self._svd_num(mi)
        vecf = np.zeros((self.nfreq, self.ntel), dtype=np.complex128)
        if np.all(svec == 0):
            return vecf.reshape(self.nfreq, 2, tel.npairs)
        beam = self.beam_ut(mi)
        for fi in self._svd_freq_iter(mi):
            noise = np.asarray(tel.noisepower(np.arange(tel.npairs), fi)).flatten()
            noise = np.concatenate([noise, noise])
            fbeam = beam[fi, : svnum[fi], :]
            vecf[fi, :] = noise * _device_gemm(fbeam.T.conj(), svec[svbounds[fi] : svbounds[fi + 1]])
        return vecf.reshape(self.nfreq, 2, tel.npairs)

    def project_vector_sky_to_svd(self, mi, vec, temponly=False):
        """[nfreq, npol, lmax+1, ...] -> [nsvd, ...] (beamtransfer.py:1324-1364)."""
        npol = 1 if temponly else self.telescope.num_pol_sky
        svnum, svbounds = self._svd_num(mi)
        vec = np.asarray(vec)
        vecf = np.zeros((svbounds[-1],) + vec.shape[3:], dtype=np.complex128)
        if np.all(vec == 0):
            return vecf
        beam = self.beam_svd(mi)
        for pi in range(npol):
            for fi in self._svd_freq_iter(mi):
                vecf[svbounds[fi] : svbounds[fi + 1]] += _device_gemm(beam[fi, : svnum[fi], pi, :], vec[fi, pi])
        return vecf

    def project_vector_svd_to_sky(self, mi, vec, temponly=False, conj=False):
        """[nsvd, ...] -> [nfreq, npol, lmax+1, ...] (beamtransfer.py:1366-1421)."""
        tel = self.telescope
        npol = 1 if temponly else tel.num_pol_sky
        svnum, svbounds = self._svd_num(mi)
        vec = np.asarray(vec)
        vecf = np.zeros((self.nfreq, tel.num_pol_sky, tel.lmax + 1) + vec.shape[1:], dtype=np.complex128)
        if np.all(vec == 0):
            return vecf
        beam = self.beam_svd(mi) if conj else self.invbeam_svd(mi)
        for pi in range(npol):
            for fi in self._svd_freq_iter(mi):
                fbeam = beam[fi, : svnum[fi], pi, :].T.conj() if conj else beam[fi, pi, :, : svnum[fi]]
                vecf[fi, pi] += _device_gemm(fbeam, vec[svbounds[fi] : svbounds[fi + 1]])
        return vecf


class BeamTransferFullSVD(BeamTransfer):
    """One SVD of the full (all sky polarisations) noise-weighted beam per (m, frequency) instead of
    the three-stage chain (beamtransfer.py:1595-1733): the same device chain run on the blocks with
    the (pol, l) axes flattened into one, i.e. SVD3 alone on an (ntel x npol*(lmax+1)) matrix.
    (The reference keeps rows whose singular value is exactly zero; they are null rows here.)"""

    @property
    def svd_len(self):
        return min((self.telescope.lmax + 1) * self.telescope.num_pol_sky, self.ntel)

    def svd_device(self, beam_blocks, skip_svd_inv=False, ms=None):
        ctx = get_context()
        nblk, F = int(beam_blocks.shape[0]), int(beam_blocks.shape[1])
        T, P, L = self.ntel, self.telescope.num_pol_sky, self.telescope.lmax + 1
        nw = self._noisew_device()
        res = ctx.svd_chain(beam_blocks.reshape(nblk, F, T, 1, P * L), nw, self.polsvcut, skip_svd_inv=skip_svd_inv)
        K = int(res["beam_svd"].shape[2])
        res["beam_svd"] = res["beam_svd"].reshape(nblk, F, K, P, L)
        if res.get("invbeam_svd") is not None:
            res["invbeam_svd"] = res["invbeam_svd"].reshape(nblk, F, P, L, K)
        return res


class BeamTransferTempSVD(BeamTransfer):
    """The old temperature-only compression (beamtransfer.py:1458-1593): ONE economy SVD of the noise-weighted
    temperature block per (m, frequency); its left vectors project every polarisation (`beam_svd = U^H w B`), no
    polarisation null space is removed.  On the device: one-sided block-Jacobi on `[w B | I_T]` with the Gram columns
    restricted to the temperature block — the accumulated row mixing is U^H, the polarised columns and the identity
    ride along as passengers — then the pseudo-inverse of the full `beam_svd` as in `invbeam_m` (a second pass on
    `[beam_svd | I]` and one grouped ZGEMM with the 1/sigma^2 weights; scipy's pinv cut rtol = max(M, N) eps)."""

    def svd_device(self, beam_blocks, skip_svd_inv=False, ms=None):
        ctx = get_context()
        nblk, F = int(beam_blocks.shape[0]), int(beam_blocks.shape[1])
        T, P, L, K = self.ntel, self.telescope.num_pol_sky, self.telescope.lmax + 1, self.svd_len
        S = P * L
        # the augmented matrices are assembled on the host (this variant is off the hot path; torch only moves memory)
        nwh = self._noisew()                                                # (F, T)
        Zh = np.zeros((nblk, F, T, S + T), dtype=np.complex128)
        Zh[..., :S] = beam_blocks.cpu().numpy().reshape(nblk, F, T, S) * nwh[None, :, :, None]
        Zh[..., S:] = np.eye(T)
        Z = ctx.to_device(Zh)
        sigma, sweeps = ctx.jacobi_rows(Z, T, S + T, 0, L, S + T, stride=T * (S + T), batch=nblk * F)   # Gram over pol 0
        ctx.sync()
        Zh = Z.cpu().numpy()
        sgh = sigma.cpu().numpy().reshape(nblk, F, T)[..., :K]
        bsvd_h = np.ascontiguousarray(Zh[:, :, :K, :S])                     # U^H (w B): rows sorted by descending sigma
        but_h = np.ascontiguousarray(Zh[:, :, :K, S:] * nwh[None, :, None, :])   # U^H diag(w)  (beamtransfer.py:1562)
        out = dict(beam_svd=ctx.to_device(bsvd_h).reshape(nblk, F, K, P, L), beam_ut=ctx.to_device(but_h),
                   singularvalues=ctx.to_device(np.ascontiguousarray(sgh)), invbeam_svd=None, sweeps=[0, 0, sweeps, 0])
        out["nmodes"] = (sgh > 0).sum(axis=-1)
        if not skip_svd_inv:
            # pinv of the (K x S) projected beam: rows orthogonalised over ALL sky columns, pinv = (S V^H)^H S^-2 W
            Yh = np.zeros((nblk, F, K, S + K), dtype=np.complex128)
            Yh[..., :S] = bsvd_h
            Yh[..., S:] = np.eye(K)
            Y = ctx.to_device(Yh)
            s2, sw2 = ctx.jacobi_rows(Y, K, S + K, 0, S, S + K, stride=K * (S + K), batch=nblk * F)
            ctx.sync()
            s2h = s2.reshape(nblk * F, K).cpu().numpy()
            smax = s2h.max(axis=1, keepdims=True)
            cut = max(K, S) * np.finfo(np.float64).eps * smax
            wts = np.where((s2h > cut) & (s2h > 0.0), 1.0 / np.where(s2h > 0.0, s2h, 1.0) ** 2, 0.0)
            dW = ctx.to_device(np.ascontiguousarray(wts))
            ib = ctx.empty((nblk * F, S, K), np.complex128)
            Yf = Y.reshape(nblk * F, K, S + K)
            ctx.zgemm(Yf, Yf[:, :, S:], ib, S, K, K, rsA=1, csA=S + K, rsB=S + K, csB=1, ldc=K, conjA=True, kscale=dW,
                      batch=nblk * F, strideA=K * (S + K), strideB=K * (S + K), strideC=S * K, stride_kscale=K)
            out["invbeam_svd"] = ib.reshape(nblk, F, P, L, K)
            out["sweeps"][3] = sw2
        return out


class BeamTransferNoSVD(BeamTransfer):
    """No SVD compression: the "SVD basis" is the telescope basis itself, ndof = nfreq * ntel
    (beamtransfer.py:1736-1968).  The batched device paths of the KL transform see the
    un-compressed beam as `beam_svd` and an identity as `beam_ut`."""

    svcut = 0.0
    noise_weight = False

    def _svd_num(self, mi):
        svnum = (np.ones(self.nfreq) * self.ntel).astype(int)
        return svnum, np.cumsum(np.insert(svnum, 0, 0))

    def _generate_svdfiles(self, regen=False, skip_svd_inv=False):
        logger.info("======== Skipping telescope SVD step ========")

    def _dev_products(self, mi):
        ctx = get_context()
        if mi not in self._dev:
            tel = self.telescope
            F, T = self.nfreq, self.ntel
            beam = np.ascontiguousarray(self.beam_m(mi).reshape(F, T, tel.num_pol_sky, tel.lmax + 1))
            eye = np.ascontiguousarray(np.broadcast_to(np.eye(T, dtype=np.complex128), (F, T, T)))
            self._dev[mi] = dict(beam_svd=ctx.to_device(beam), beam_ut=ctx.to_device(eye), singularvalues=None)
        return self._dev[mi]

    def project_matrix_sky_to_svd(self, mi, mat, temponly=False):
        return self.project_matrix_sky_to_telescope(mi, mat, temponly=temponly).reshape(self.ndof(mi), self.ndof(mi))

    def project_vector_sky_to_svd(self, mi, vec, *args, **kwargs):
        return self.project_vector_sky_to_telescope(mi, vec).flatten()

    def project_matrix_telescope_to_svd(self, mi, mat):
        return np.asarray(mat).reshape(self.ndof(mi), self.ndof(mi))

    def project_matrix_diagonal_telescope_to_svd(self, mi, dmat, *args, **kwargs):
        return np.diag(np.asarray(dmat).flatten())

    def project_vector_telescope_to_svd(self, mi, vec, *args, **kwargs):
        return np.asarray(vec).flatten()

    def project_vector_svd_to_sky(self, mi, vec, temponly=False, conj=False):
        if temponly:
            raise NotImplementedError("temponly not implemented for no-SVD project_vector_svd_to_sky!")
        tel = self.telescope
        vec = np.asarray(vec)
        svec = np.zeros((self.nfreq, tel.num_pol_sky, tel.lmax + 1) + vec.shape[1:], dtype=np.complex128)
        v = vec.reshape(self.nfreq, self.ntel, -1)
        if conj:
            mats = self.beam_m(mi).reshape(self.nfreq, self.ntel, self.nsky).transpose(0, 2, 1).conj()
        else:
            mats = self.invbeam_m(mi).reshape(self.nfreq, self.nsky, self.ntel)
        for fi in range(self.nfreq):
            svec[fi] = _device_gemm(mats[fi], v[fi]).reshape((tel.num_pol_sky, tel.lmax + 1) + vec.shape[1:])
        return svec

    def beam_svd(self, mi, *args, **kwargs):
        return self.beam_m(mi)

    def ndof(self, mi, *args, **kwargs):
        return self.ntel * self.nfreq

    @property
    def ndofmax(self):
        return self.ntel * self.nfreq


def _device_gemm(A, B):
    """A @ B through the grouped ZGEMM of libdriftmi (small helper for the vector projections)."""
    ctx = get_context()
    A = np.ascontiguousarray(A, dtype=np.complex128)
    Bm = np.asarray(B, dtype=np.complex128)
    vec = Bm.ndim == 1
    B2 = np.ascontiguousarray(Bm.reshape(Bm.shape[0], -1))
    M, K = A.shape
    N = B2.shape[1]
    dC = ctx.empty((M, max(N, 1)), np.complex128)
    if M and N:
        ctx.zgemm(ctx.to_device(A), ctx.to_device(B2), dC, M, N, K, rsA=K, csA=1, rsB=N, csB=1, ldc=N)
        ctx.sync()
    out = dC.cpu().numpy()[:, :N]
    return out[:, 0] if vec else out.reshape((M,) + Bm.shape[1:])


def _device_bgemv(mats, vecs):
    """out[b] = mats[b] @ vecs[b] for a stack of matrices (one strided-batched ZGEMM)."""
    ctx = get_context()
    nb, M, K = mats.shape
    dA = ctx.to_device(np.ascontiguousarray(mats, dtype=np.complex128))
    dB = ctx.to_device(np.ascontiguousarray(vecs, dtype=np.complex128).reshape(nb, K, 1))
    dC = ctx.empty((nb, M, 1), np.complex128)
    ctx.zgemm(dA, dB, dC, M, 1, K, rsA=K, csA=1, rsB=1, csB=1, ldc=1, batch=nb, strideA=M * K, strideB=K, strideC=M)
    ctx.sync()
    return dC.cpu().numpy().reshape(nb, M)
