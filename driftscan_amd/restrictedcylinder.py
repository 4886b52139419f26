"""Cylinders whose beam is cut down in declination by a box or a Gaussian window centred on the zenith
(drift/telescope/restrictedcylinder.py).  The window is host work on ``self._angpos``; the cylinder pattern
under it comes from the device kernel (``CylinderTelescope._pattern_host``)."""
import numpy as np

from . import config, cylinder


def gaussian_fwhm(x, fwhm):
    sigma = fwhm / (8.0 * np.log(2.0)) ** 0.5
    return np.exp(-(x**2) / (2.0 * sigma**2))


class RestrictedBeam(cylinder.CylinderTelescope):
    beam_height = config.Property(proptype=float, default=30.0)   # degrees
    beam_type = config.Property(proptype=str, default="box")

    def beam_spec(self, beamclass, freq_index):
        return None   # the windowed beams are assembled on the host

    def _dtheta(self):
        d = self._angpos - np.asarray(self.zenith)[np.newaxis, :]
        d = np.where((d[:, 1] < np.pi)[:, np.newaxis], d, d - np.array([0.0, 2.0 * np.pi])[np.newaxis, :])
        return np.abs(d[:, 0])

    def bmask_gaussian(self, feed, freq):
        return gaussian_fwhm(self._dtheta(), np.radians(self.beam_height))

    def bmask_box(self, feed, freq):
        return np.abs(self._dtheta() / np.radians(self.beam_height)) < 0.5

    def _window(self, feed, freq):
        return {"gaussian": self.bmask_gaussian, "box": self.bmask_box}[self.beam_type](feed, freq)


class RestrictedCylinder(RestrictedBeam, cylinder.UnpolarisedCylinderTelescope):
    def beam(self, feed, freq):
        return self._window(feed, freq) * cylinder.UnpolarisedCylinderTelescope.beam(self, feed, freq)


class RestrictedPolarisedCylinder(RestrictedBeam, cylinder.PolarisedCylinderTelescope):
    def beamx(self, feed, freq):
        return self._window(feed, freq)[:, np.newaxis] * cylinder.PolarisedCylinderTelescope.beamx(self, feed, freq)

    def beamy(self, feed, freq):
        return self._window(feed, freq)[:, np.newaxis] * cylinder.PolarisedCylinderTelescope.beamy(self, feed, freq)


class RestrictedExtra(RestrictedCylinder):
    """Extra feeds at given N-S positions in front of the regular ones of every cylinder."""

    extra_feeds = config.Property(proptype=np.array, default=[])

    def feed_positions_cylinder(self, cylinder_index):
        pos = super(RestrictedExtra, self).feed_positions_cylinder(cylinder_index)
        extra = np.asarray(self.extra_feeds, dtype=np.float64).reshape(-1)
        head = np.stack([np.full(extra.size, cylinder_index * self.cylinder_spacing), extra], axis=1)
        return np.concatenate([head, pos])
