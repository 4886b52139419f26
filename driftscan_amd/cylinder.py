"""Regular cylinder arrays (drift/telescope/cylinder.py) and the host part of their
beam model (drift/telescope/cylbeam.py): feed grid, in-cylinder masking, and the
knots of the E-W Fraunhofer pattern spline.  Pixel-level evaluation is on the GPU
(``dm_bt_beam_cyl``)."""
import numpy as np

from . import config, telescope


def beam_exptan_host(sintheta, fwhm):
    """exp(-alpha s^2/(1 - s^2 + 1e-100)) on a handful of host samples (illumination of the FFT)."""
    alpha = np.log(2.0) / (2.0 * np.tan(fwhm / 2.0) ** 2)
    s2 = np.asarray(sintheta) ** 2
    return np.exp(-alpha * s2 / (1.0 - s2 + 1e-100))


def natural_spline_y2(x, y):
    """Second derivatives of the natural cubic spline (what cora's Interpolater builds)."""
    n = x.size
    y2 = np.zeros(n)
    u = np.zeros(n)
    for i in range(1, n - 1):
        sig = (x[i] - x[i - 1]) / (x[i + 1] - x[i - 1])
        p = sig * y2[i - 1] + 2.0
        y2[i] = (sig - 1.0) / p
        u[i] = (y[i + 1] - y[i]) / (x[i + 1] - x[i]) - (y[i] - y[i - 1]) / (x[i] - x[i - 1])
        u[i] = (6.0 * u[i] / (x[i + 1] - x[i - 1]) - sig * u[i - 1]) / p
    for k in range(n - 2, -1, -1):
        y2[k] = y2[k] * y2[k + 1] + u[k]
    return y2


_pattern_cache = {}


def fraunhofer_table(fwhm, width):
    """Spline knots (kx, fx, fx'') of the 1-D Fraunhofer pattern of an exptan-illuminated
    aperture `width` wavelengths across: 512 illumination samples zero-padded to 8192,
    FFT, normalised to unit maximum, trimmed to |kx| < 1.1 (cylbeam.py:52-95)."""
    key = (float(fwhm), float(width))
    if key not in _pattern_cache:
        num, res = 512, 16
        hnum = num // 2 - 1
        ua = -1.0 * np.linspace(-1.0, 1.0, num, endpoint=False)[::-1]
        ax = beam_exptan_host(2 * ua / (1 + ua**2), fwhm)
        axe = np.zeros(res * num)
        axe[: hnum + 2] = ax[hnum:]
        axe[-hnum:] = ax[:hnum]
        fx = np.fft.fft(axe).real
        kx = 2 * np.fft.fftfreq(res * num, ua[1] - ua[0]) / width
        fx = np.fft.fftshift(fx) / fx.max()
        kx = np.fft.fftshift(kx)
        sel = np.abs(kx) < 1.1
        kx, fx = np.ascontiguousarray(kx[sel]), np.ascontiguousarray(fx[sel])
        _pattern_cache[key] = (kx, fx, natural_spline_y2(kx, fx))
        if len(_pattern_cache) > 256:
            _pattern_cache.pop(next(iter(_pattern_cache)))
    return _pattern_cache[key]


class CylinderTelescope(telescope.TransitTelescope):
    """Feeds on a regular grid along `num_cylinders` N-S cylinders."""

    num_cylinders = config.Property(proptype=int, default=2)
    num_feeds = config.Property(proptype=int, default=6)
    cylinder_width = config.Property(proptype=float, default=20.0)
    feed_spacing = config.Property(proptype=float, default=0.5)
    in_cylinder = config.Property(proptype=config.truthy, default=True)
    touching = config.Property(proptype=config.truthy, default=True)
    cylspacing = config.Property(proptype=float, default=0.0)
    non_commensurate = config.Property(proptype=config.truthy, default=False)
    e_width = config.Property(proptype=float, default=0.7)
    h_width = config.Property(proptype=float, default=1.0)

    _fwhm_e = 2.0 * np.pi / 3.0
    _fwhm_h = 2.0 * np.pi / 3.0

    @property
    def fwhm_e(self):
        return self._fwhm_e * self.e_width

    @property
    def fwhm_h(self):
        return self._fwhm_h * self.h_width

    @property
    def u_width(self):
        return self.cylinder_width

    @property
    def v_width(self):
        return 0.0

    def _unique_baselines(self):
        base_map, base_mask = super(CylinderTelescope, self)._unique_baselines()
        if not self.in_cylinder:
            d, _ = self._separations()
            base_mask = base_mask & (d[..., 0] != 0.0)  # drop pairs on the same cylinder (cylinder.py:72-109)
            base_map = telescope._label_keys(base_map, base_mask)
        return base_map, base_mask

    @property
    def cylinder_spacing(self):
        return self.cylinder_width if self.touching else self.cylspacing

    def feed_positions_cylinder(self, cylinder_index):
        if cylinder_index >= self.num_cylinders or cylinder_index < 0:
            raise Exception("Cylinder index is invalid.")
        nf, sp = self.num_feeds, self.feed_spacing
        if self.non_commensurate:
            nf = self.num_feeds - cylinder_index
            sp = self.feed_spacing / (nf - 1.0) * nf
        pos = np.empty([nf, 2], dtype=np.float64)
        pos[:, 0] = cylinder_index * self.cylinder_spacing
        pos[:, 1] = np.arange(nf) * sp
        return pos

    @property
    def _single_feedpositions(self):
        return np.vstack([self.feed_positions_cylinder(i) for i in range(self.num_cylinders)])

    # ---- beam description consumed by btgen ------------------------------------
    def beam_spec(self, beamclass, freq_index):
        """(kind, spline table, fwhm_ns) of the field pattern of `beamclass` at a frequency.

        kind 0 = unpolarised amplitude, 1 = X dipole, 2 = Y dipole.  X uses the E-plane
        width E-W and the H-plane width N-S; Y swaps them (cylbeam.py:150-212)."""
        raise NotImplementedError

    def _pattern_host(self, kind, fwhm_ew, fwhm_ns, freq_index):
        """The cylinder field pattern on the pixels of ``self._nside`` as a host array — (npix,) for kind 0,
        (npix, 2) otherwise — evaluated by the device kernel (``dm_bt_beam_cyl``) and copied back: what the
        reference's ``beam`` / ``beamx`` / ``beamy`` return (cylinder.py:171-218), times the horizon mask.
        For the subclasses that modify the cylinder beam on the host (restricted / perturbed cylinders)."""
        from . import btgen, healpix
        from .device import get_context

        if self._nside is None:
            raise RuntimeError("call _init_trans(nside) first")
        ctx = get_context()
        nside = int(self._nside)
        cth, sth = healpix.ring_trig(nside)
        ncomp = 1 if kind == 0 else 2
        out = ctx.empty((healpix.npix(nside) * ncomp,), np.float64)
        width = self.cylinder_width / self.wavelengths[freq_index]
        ctx.bt_beam_cyl(nside, cth, sth, btgen.telescope_frame(self.zenith), kind, fraunhofer_table(fwhm_ew, width),
                        fwhm_ns, out)
        h = out.cpu().numpy()
        return h if kind == 0 else h.reshape(-1, 2)


class UnpolarisedCylinderTelescope(CylinderTelescope, telescope.SimpleUnpolarisedTelescope):
    def beam_spec(self, beamclass, freq_index):
        width = self.cylinder_width / self.wavelengths[freq_index]
        return 0, fraunhofer_table(self.fwhm_h, width), self.fwhm_h  # cylinder.py:171-194

    def beam(self, feed, freq):
        return self._pattern_host(0, self.fwhm_h, self.fwhm_h, freq)


class PolarisedCylinderTelescope(CylinderTelescope, telescope.SimplePolarisedTelescope):
    def beam_spec(self, beamclass, freq_index):
        width = self.cylinder_width / self.wavelengths[freq_index]
        if beamclass % 2 == 0:  # X feed: beam_x(width, fwhm_e, fwhm_h)
            return 1, fraunhofer_table(self.fwhm_e, width), self.fwhm_h
        return 2, fraunhofer_table(self.fwhm_h, width), self.fwhm_e  # Y feed: widths swapped

    def beamx(self, feed, freq):
        return self._pattern_host(1, self.fwhm_e, self.fwhm_h, freq)

    def beamy(self, feed, freq):
        return self._pattern_host(2, self.fwhm_h, self.fwhm_e, freq)
