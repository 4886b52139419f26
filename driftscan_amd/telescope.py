"""Transit-telescope description: frequencies, feed pairs -> unique baselines,
redundancy, band limits, noise, and the device-side transfer-matrix generation.

Mirrors the attribute surface that ``BeamTransfer`` / ``KLTransform`` read from the
reference's ``drift.core.telescope.TransitTelescope`` (drift/core/telescope.py):
``nfreq, nbase, npairs, num_pol_sky, lmax, mmax, frequencies, wavelengths,
baselines, uniquepairs, redundancy, feedmap, feedmask, feedconj, included_freq,
included_baseline, included_pol, noisepower(bl, f), tsys_flat, zenith``.

The geometry is host work (a few thousand feed pairs); the per-pixel work of
``transfer_matrices`` runs on the GPU through ``driftscan_amd.btgen``.
"""
import numpy as np

from . import config

# Physical constants as in cora.util.units (absent from this image)
SPEED_OF_LIGHT = 299792458.0
T_SIDEREAL = 23.9344696 * 3600.0


def max_lm(baselines, wavelengths, uwidth, vwidth=0.0):
    """Largest (l, m) a baseline is sensitive to (drift/core/telescope.py:99-122):
    m = ceil(2 pi (|u| + w_u) / lambda), l = ceil(sqrt(m^2 + (2 pi (|v| + w_v) / lambda)^2))."""
    umax = (np.abs(baselines[:, 0]) + uwidth) / wavelengths
    vmax = (np.abs(baselines[:, 1]) + vwidth) / wavelengths
    mmax = np.ceil(2 * np.pi * umax).astype(np.int64)
    lmax = np.ceil((mmax**2 + (2 * np.pi * vmax) ** 2) ** 0.5).astype(np.int64)
    return lmax, mmax


def sph_to_cart(sph):
    """Unit vectors of (theta, phi) positions (cora.util.coord.sph_to_cart on two-column input)."""
    sph = np.asarray(sph, dtype=np.float64)
    st = np.sin(sph[..., 0])
    return np.stack([st * np.cos(sph[..., 1]), st * np.sin(sph[..., 1]), np.cos(sph[..., 0])], axis=-1)


def sph_dot(a, b):
    """Cosine of the angle between (theta, phi) positions (cora.util.coord.sph_dot)."""
    return np.sum(sph_to_cart(a) * sph_to_cart(b), axis=-1)


def _label_keys(keys, mask):
    """Integer labels 0.. of the distinct values of `keys` among the entries selected by `mask` (in sorted
    order of the values: complex keys sort on the real part first), -1 elsewhere (telescope.py:66-79)."""
    keys = np.asarray(keys)
    out = np.full(keys.shape, -1, dtype=np.int64)
    sel = np.nonzero(np.ones(keys.shape, dtype=bool) if mask is None else mask)
    if sel[0].size:
        out[sel] = np.asarray(np.unique(keys[sel], return_inverse=True)[1]).reshape(-1)
    return out


def _representatives(labels, mask, ngrp):
    """(i, j) of the first pair in row-major order carrying each label among `mask` (telescope.py:82-96)."""
    flat = np.nonzero(mask.ravel())[0]
    _, first = np.unique(labels.ravel()[flat], return_index=True)
    if first.size != ngrp:
        raise ValueError("a baseline group has no member with the stored orientation")
    rep = flat[first]
    return rep // labels.shape[1], rep % labels.shape[1]


class TransitTelescope(config.Reader):
    """Base class; subclasses provide ``feedpositions``, ``beamclass``, ``u_width``,
    ``v_width`` and the beam description consumed by ``btgen``."""

    freq_lower = config.Property(proptype=float, default=None)
    freq_upper = config.Property(proptype=float, default=None)
    freq_start = config.Property(proptype=float, default=800.0)
    freq_end = config.Property(proptype=float, default=400.0)
    num_freq = config.Property(proptype=int, default=1024)
    freq_mode = config.enum(["centre", "centre_nyquist", "edge"], default="centre")
    channel_bin = config.Property(proptype=int, default=1)
    channel_range = config.Property(proptype=list)
    channel_list = config.Property(proptype=list)

    tsys_flat = config.Property(proptype=float, default=50.0, key="tsys")
    ndays = config.Property(proptype=int, default=733)

    accuracy_boost = config.Property(proptype=float, default=1.0)
    l_boost = config.Property(proptype=float, default=1.0)
    # The two settings of healpy.map2alm the reference reaches through cora.util.hputil.sphtrans_complex[_pol]
    # (telescope.py:1179-1191, :1288-1312; neither package can be read here): `iter`, Jacobi refinements of the quadrature
    # — healpy's documented default is 3 and nothing in the reference overrides it, so 3 is the default here (iter = 0
    # moves beam_m by up to 6e-3 of the block scale on tests/testparams.yaml, far outside the reference's own
    # approx(rel=1e-4)) — and optional per-ring weight factors {nside: array(4 nside - 1)}
    sht_iter = config.Property(proptype=int, default=3)
    sht_ring_weights = None
    force_lmax = config.Property(proptype=int, default=None)
    force_mmax = config.Property(proptype=int, default=None)

    minlength = config.Property(proptype=float, default=0.0)
    maxlength = config.Property(proptype=float, default=1.0e7)
    auto_correlations = config.Property(proptype=config.truthy, default=False)
    local_origin = config.Property(proptype=config.truthy, default=True)

    skip_freq = config.list_type(type_=int, default=[])
    skip_baselines = config.list_type(type_=int, default=[])

    _bl_tol = 6  # decimals kept when comparing separations (telescope.py:554)

    def __getstate__(self):
        """The telescope pickle (bt/telescopeobject.pickle, beamtransfer.py:199-202) holds the description, not the
        per-process memo of BT-gen band limits."""
        state = dict(self.__dict__)
        state.pop("_btgen_memo", None)
        state.pop("_btgen_chunk_memo", None)
        return state
    _npol_sky_ = 1

    def __init__(self, latitude=45, longitude=0, **kwargs):
        self.latitude = latitude
        self.longitude = longitude
        self._pairs = None
        self._frequencies = None

    def _finalise_config(self):
        self._pairs = None
        self._frequencies = None

    # ---- pointing -------------------------------------------------------
    @property
    def zenith(self):
        """[theta, phi] of the zenith (telescope.py:268-291)."""
        theta = np.pi / 2.0 - np.radians(self.latitude)
        phi = 0.0 if self.local_origin else np.remainder(np.radians(self.longitude), 2 * np.pi)
        return np.array([theta, phi])

    # ---- frequencies ----------------------------------------------------
    @property
    def frequencies(self):
        if self._frequencies is None:
            self.calculate_frequencies()
        return self._frequencies

    def calculate_frequencies(self):
        """Channel centres in MHz (telescope.py:386-431)."""
        if self.freq_lower or self.freq_upper:
            self.freq_start, self.freq_end = self.freq_lower, self.freq_upper
        n = self.num_freq
        if self.freq_mode == "centre":
            freq = np.linspace(self.freq_start, self.freq_end, n, endpoint=False)
        elif self.freq_mode == "centre_nyquist":
            freq = np.linspace(self.freq_start, self.freq_end, n, endpoint=True)
        else:
            df = abs(self.freq_end - self.freq_start) / n
            freq = self.freq_start + df * (np.arange(n) + 0.5)
        if self.channel_bin > 1:
            if n % self.channel_bin != 0:
                raise ValueError("Channel binning must exactly divide the total number of channels")
            freq = freq.reshape(-1, self.channel_bin).mean(axis=1)
        if self.channel_list is not None:
            raise NotImplementedError("`channel_list` is not yet supported")
        if self.channel_range is not None:
            freq = freq[self.channel_range[0] : self.channel_range[1]]
        self._frequencies = freq

    @property
    def wavelengths(self):
        return SPEED_OF_LIGHT / (1e6 * self.frequencies)

    @property
    def nfreq(self):
        return self.frequencies.shape[0]

    # ---- feeds and baselines ---------------------------------------------
    @property
    def nfeed(self):
        return self.feedpositions.shape[0]

    @property
    def num_pol_sky(self):
        return self._npol_sky_

    def _separations(self):
        """(d, blen): separation vectors and lengths of every ordered feed pair."""
        pos = self.feedpositions
        d = pos[:, None, :] - pos[None, :, :]
        return d, np.sqrt(np.sum(d**2, axis=-1))

    # The three hooks below are the reference's plug-in points for telescope classes
    # (telescope.py:556-626): integer label maps over the (nfeed, nfeed) pair grid plus masks.
    def _unique_baselines(self):
        """(label map of equal separations, mask of the pairs taking part)."""
        d, blen = self._separations()
        key = np.around(d[..., 0] + 1.0j * d[..., 1], self._bl_tol)
        mask = (blen >= self.minlength) & (blen <= self.maxlength)
        if not self.auto_correlations:
            mask &= blen > 0.0
        return _label_keys(key, mask), mask

    def _unique_beams(self):
        """(label map of equal (class_i, class_j), mask: everything but the diagonal unless auto_correlations)."""
        cls = np.asarray(self.beamclass)
        lab = _label_keys(cls[:, None] + 1.0j * cls[None, :], None)
        mask = np.ones((self.nfeed, self.nfeed), dtype=bool) if self.auto_correlations else ~np.eye(self.nfeed, dtype=bool)
        return lab, mask

    def _get_unique(self):
        """(feedmap, feedmask, feedconj) before orientation and ordering: pairs with the same separation and
        the same beam classes are one baseline, a pair and its transpose share the label and one of the two
        is marked as the conjugate."""
        base, bmask = self._unique_baselines()
        beam, cmask = self._unique_beams()
        mask = bmask & cmask
        comb = _label_keys(base + 1.0j * beam, mask)
        conj = comb > comb.T
        comb = _label_keys(np.minimum(comb, comb.T), mask)
        return comb, mask, conj

    def calculate_feedpairs(self):
        """Group feed pairs into unique baselines (telescope.py:507-675).

        ``_get_unique`` labels the pairs; the stored orientation of a baseline is then made to point East (or
        due North), baselines are ordered lexicographically in (u, v, class_j, class_i) of their representative
        pair — the first member with the stored orientation in row-major (i, j) order — and the redundancy
        counts the members with the stored orientation.
        """
        pos = self.feedpositions
        cls = np.asarray(self.beamclass)
        feedmap, mask, conj = self._get_unique()
        feedmap = np.array(feedmap, dtype=np.int64)
        mask = np.asarray(mask, dtype=bool)
        conj = np.asarray(conj, dtype=bool)
        ngrp = int(feedmap[mask].max()) + 1 if mask.any() else 0

        # orientation: flip the groups whose representative points West (or due South)
        ri, rj = _representatives(feedmap, mask & ~conj, ngrp)
        sep = pos[ri] - pos[rj]
        west = (sep[:, 0] < 0.0) | ((sep[:, 0] == 0.0) & (sep[:, 1] < 0.0))
        flip = np.zeros(feedmap.shape, dtype=bool)
        flip[mask] = west[feedmap[mask]]
        conj = conj ^ flip

        # order on the unrounded separation of the (new) representative, then (class_j, class_i)
        ri, rj = _representatives(feedmap, mask & ~conj, ngrp)
        sort_arr = np.zeros(ngrp, dtype=np.dtype("f8,f8,i4,i4"))
        sort_arr["f0"], sort_arr["f1"] = pos[ri, 0] - pos[rj, 0], pos[ri, 1] - pos[rj, 1]
        sort_arr["f2"], sort_arr["f3"] = cls[rj], cls[ri]
        perm = np.argsort(sort_arr)
        newlabel = np.empty(ngrp, dtype=np.int64)
        newlabel[perm] = np.arange(ngrp)
        feedmap[mask] = newlabel[feedmap[mask]]

        uniquepairs = np.stack([ri[perm], rj[perm]], axis=1).reshape(-1, 2)
        self._pairs = dict(
            feedmap=feedmap,
            feedmask=mask,
            feedconj=conj,
            uniquepairs=uniquepairs,
            redundancy=np.bincount(feedmap[mask & ~conj], minlength=ngrp),
            baselines=pos[uniquepairs[:, 0]] - pos[uniquepairs[:, 1]],
        )

    def _pair(self, key):
        if self._pairs is None:
            self.calculate_feedpairs()
        return self._pairs[key]

    baselines = property(lambda self: self._pair("baselines"))
    redundancy = property(lambda self: self._pair("redundancy"))
    uniquepairs = property(lambda self: self._pair("uniquepairs"))
    feedmap = property(lambda self: self._pair("feedmap"))
    feedmask = property(lambda self: self._pair("feedmask"))
    feedconj = property(lambda self: self._pair("feedconj"))

    @property
    def npairs(self):
        return self.uniquepairs.shape[0]

    nbase = npairs

    # ---- harmonic band limits ---------------------------------------------
    @property
    def lmax(self):
        if self.force_lmax is not None:
            return self.force_lmax
        l, m = max_lm(self.baselines, self.wavelengths.min(), self.u_width, self.v_width)
        return int(np.ceil(l.max() * self.l_boost))

    @property
    def mmax(self):
        if self.force_mmax is not None:
            return self.force_mmax
        l, m = max_lm(self.baselines, self.wavelengths.min(), self.u_width, self.v_width)
        return int(np.ceil(m.max() * self.l_boost))

    def baseline_lmax(self, bl_indices, f_indices):
        """Per-(baseline, frequency) band limit used by transfer_matrices (telescope.py:792-802)."""
        bl_indices = np.asarray(bl_indices).reshape(-1)
        f_indices = np.asarray(f_indices).reshape(-1)
        l, m = max_lm(self.baselines[bl_indices], self.wavelengths[f_indices], self.u_width, self.v_width)
        return np.ceil(self.l_boost * l).astype(np.int64), np.ceil(self.l_boost * m).astype(np.int64)

    # ---- skipping -----------------------------------------------------------
    @property
    def included_freq(self):
        return np.array([i for i in range(self.nfreq) if i not in self.skip_freq], dtype=int)

    @property
    def included_baseline(self):
        return np.array([i for i in range(self.nbase) if i not in self.skip_baselines], dtype=int)

    @property
    def included_pol(self):
        return np.arange(self.num_pol_sky)

    # ---- noise ----------------------------------------------------------------
    def tsys(self, f_indices=None):
        freq = self.frequencies if f_indices is None else self.frequencies[f_indices]
        return np.ones_like(freq) * self.tsys_flat

    def noisepower(self, bl_indices, f_indices, ndays=None):
        """White noise power per m-mode, Tsys^2 / (2 pi dnu t_sid/(2 pi) ndays) / redundancy
        (telescope.py:894-926)."""
        ndays = self.ndays if not ndays else ndays
        bl_indices, f_indices = np.broadcast_arrays(bl_indices, f_indices)
        bw = np.abs(self.frequencies[1] - self.frequencies[0]) * 1e6
        delnu = T_SIDEREAL * bw / (2 * np.pi)
        noisepower = self.tsys(f_indices) ** 2 / (2 * np.pi * delnu * ndays)
        return noisepower / self.redundancy[bl_indices]

    # ---- beams --------------------------------------------------------------------
    # A telescope class describes its primary beams in one of two ways:
    #   * ``beam_spec(beamclass, freq)`` -> (kind, spline table, fwhm_ns): evaluated per pixel ON THE DEVICE
    #     (``dm_bt_beam_cyl``; the cylinder classes), or
    #   * the reference's plug-in interface (telescope.py:943-952, :1136, :1399-1447): ``beam(feed, freq)``
    #     (``beamx`` / ``beamy`` for polarised classes) returning the field pattern on ``self._angpos`` as a host
    #     array — [npix] amplitudes, or [npix, 2] (theta, phi) components.  ``btgen`` uploads one such map per
    #     (frequency, beam class) and the device does the rest.
    _nside = None
    _angpos = None
    _horizon = None
    complex_beams = False   # True: beam() may return complex patterns (two-call BT-gen path, dm_bt_maps_c)

    def beam_spec(self, beamclass, freq_index):
        return None

    def _init_trans(self, nside):
        """Pixel centres and horizon mask of a HEALPix resolution (telescope.py:943-952)."""
        from . import healpix

        self._nside = int(nside)
        self._angpos = healpix.ang_positions(self._nside)
        # drift/core/visibility.py:27-46: signbit(-n . zenith), the hemisphere above the horizon
        self._horizon = np.signbit(-sph_dot(self._angpos, self.zenith))

    def beam(self, feed, freq):
        raise NotImplementedError("%s defines neither beam_spec() nor beam()" % type(self).__name__)

    def _beam_host(self, feed, freq_index, nside):
        """Field pattern of `feed` at a frequency on the pixels of `nside`, zero below the horizon: (npix,) for an
        unpolarised telescope, (npix, 2) for a polarised one; float64, or complex128 if the class returns a pattern
        with a non-zero imaginary part."""
        if self._nside != int(nside) or self._angpos is None:
            self._init_trans(nside)
        b = np.asarray(self.beam(feed, freq_index))
        if np.iscomplexobj(b) and not np.any(b.imag != 0.0):
            b = b.real
        want = (self._angpos.shape[0],) if self.num_pol_sky == 1 else (self._angpos.shape[0], 2)
        if b.shape != want:
            raise ValueError("beam(%d, %d) returned shape %r, expected %r" % (feed, freq_index, b.shape, want))
        hz = self._horizon.astype(np.float64)
        # a genuinely complex pattern stays complex128: btgen then takes the complex-pattern kernels (dm_bt_maps_c)
        dt = np.complex128 if np.iscomplexobj(b) else np.float64
        return np.ascontiguousarray(b.astype(dt) * (hz if b.ndim == 1 else hz[:, None]))

    # ---- transfer matrices ------------------------------------------------------
    def transfer_matrices(self, bl_indices, f_indices, global_lmax=True):
        """(nfb, P, lside+1, 2*lside+1) a_lm of the requested (baseline, frequency) pairs,
        non-centred m, computed on the GPU (telescope.py:755-830)."""
        from . import btgen

        bl = np.asarray(bl_indices).reshape(-1)
        fi = np.asarray(f_indices).reshape(-1)
        bl, fi = np.broadcast_arrays(bl, fi)
        if ((bl < 0) | (bl >= self.npairs)).any():
            raise ValueError("Baseline indices aren't valid")
        if ((fi < 0) | (fi >= self.nfreq)).any():
            raise ValueError("Frequency indices aren't valid")
        return btgen.transfer_matrices(self, bl, fi, global_lmax=global_lmax)


class UnpolarisedTelescope(TransitTelescope):
    _npol_sky_ = 1

    def noisepower(self, bl_indices, f_indices, ndays=None):
        """Unpolarised telescopes carry half the noise, with a trailing axis (telescope.py:1197-1221)."""
        bnoise = TransitTelescope.noisepower(self, bl_indices, f_indices, ndays)
        return bnoise[..., np.newaxis] * 0.5


class PolarisedTelescope(TransitTelescope):
    _npol_sky_ = 4
    skip_V = config.Property(proptype=config.truthy, default=False)
    skip_pol = config.Property(proptype=config.truthy, default=False)

    @property
    def included_pol(self):
        return np.arange(1 if self.skip_pol else (3 if self.skip_V else 4))


class SimpleUnpolarisedTelescope(UnpolarisedTelescope):
    @property
    def beamclass(self):
        return np.zeros(self._single_feedpositions.shape[0], dtype=np.int64)

    @property
    def feedpositions(self):
        return self._single_feedpositions


class SimplePolarisedTelescope(PolarisedTelescope):
    @property
    def polarisation(self):
        return np.asarray(["X" if c % 2 == 0 else "Y" for c in self.beamclass], dtype=str)

    @property
    def beamclass(self):
        n = self._single_feedpositions.shape[0]
        return np.concatenate((np.zeros(n), np.ones(n))).astype(np.int64)

    @property
    def feedpositions(self):
        return np.concatenate((self._single_feedpositions, self._single_feedpositions))

    def beam(self, feed, freq):
        """Field pattern of a feed: ``beamx`` for the X feeds, ``beamy`` for the Y feeds (telescope.py:1399-1403)."""
        return self.beamx(feed, freq) if self.polarisation[feed] == "X" else self.beamy(feed, freq)

    def beamx(self, feed, freq):
        raise NotImplementedError

    def beamy(self, feed, freq):
        raise NotImplementedError
