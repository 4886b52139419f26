"""Focal-plane array: `beam_num_u x beam_num_v` beams formed on one dish, all at zero separation, each pointing
its own way (drift/telescope/focalplane.py).  Only the auto-correlation of each beam is a "baseline", so the
class needs ``auto_correlations: true``.

The reference class cannot be instantiated as shipped (it defines no ``beamclass``, an abstract property of its
base): here every beam is its own class, which is what its ``_unique_beams`` expresses.  Geometry parity for this
class is therefore UNPINNED; the beams themselves are the same closed forms.
"""
import numpy as np

from . import config, telescope
from .gmrt import gaussian_beam


class FocalPlaneArray(telescope.UnpolarisedTelescope):
    beam_num_u = config.Property(proptype=int, default=10)
    beam_num_v = config.Property(proptype=int, default=10)
    beam_spacing_u = config.Property(proptype=float, default=0.1)   # degrees
    beam_spacing_v = config.Property(proptype=float, default=0.1)
    beam_size = config.Property(proptype=float, default=0.1)        # FWHM, degrees
    beam_pivot = config.Property(proptype=float, default=400.0)     # MHz
    beam_freq_scale = config.Property(proptype=config.truthy, default=True)
    square_beam = config.Property(proptype=config.truthy, default=False)

    @property
    def beam_pointings(self):
        """(nfeed, 2) [theta, phi] of the beam centres (focalplane.py:65-82)."""
        pu = np.radians(self.beam_spacing_u * (np.arange(self.beam_num_u) - (self.beam_num_u - 1) / 2.0)) + self.zenith[1]
        pv = np.radians(self.beam_spacing_v * (np.arange(self.beam_num_v) - (self.beam_num_v - 1) / 2.0)) + self.zenith[0]
        pnt = np.zeros((self.beam_num_u, self.beam_num_v, 2))
        pnt[:, :, 1] = pu[:, np.newaxis]
        pnt[:, :, 0] = pv[np.newaxis, :]
        return pnt.reshape(-1, 2)

    def _offsets(self, feed):
        """|theta - theta0|, |phi - phi0| (wrapped) in units of the beam size (focalplane.py:97-108)."""
        d = self._angpos - self.beam_pointings[feed][np.newaxis, :]
        d = np.where((d[:, 1] < np.pi)[:, np.newaxis], d, d - np.array([0.0, 2.0 * np.pi])[np.newaxis, :])
        return np.abs(d) / np.radians(self.beam_size)

    def beam(self, feed, freq):
        if self.square_beam:
            d = self._offsets(feed)
            return np.logical_and(d[:, 0] < 0.5, d[:, 1] < 0.5).astype(np.float64)
        fwhm = self.beam_size * self.frequencies[freq] / self.beam_pivot if self.beam_freq_scale else self.beam_size
        sigma = np.radians(fwhm) / (8.0 * np.log(2.0)) ** 0.5
        return gaussian_beam(self._angpos, self.beam_pointings[feed], sigma)

    @property
    def dish_width(self):
        return telescope.SPEED_OF_LIGHT / self.beam_pivot * 1e-6 / np.radians(self.beam_size)

    @property
    def u_width(self):
        return self.dish_width

    @property
    def v_width(self):
        return self.dish_width

    @property
    def nfeed(self):
        return self.beam_num_u * self.beam_num_v

    @property
    def feedpositions(self):
        return np.zeros([self.nfeed, 2])

    @property
    def beamclass(self):
        return np.arange(self.nfeed)

    def _unique_beams(self):
        mask = np.identity(self.nfeed, dtype=bool)
        return telescope._label_keys(np.diag(np.arange(self.nfeed)), mask), mask
