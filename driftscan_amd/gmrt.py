"""GMRT-like array: dishes at tabulated positions with a Gaussian primary beam whose width scales as
150 MHz / frequency (drift/telescope/gmrt.py).  The antenna table is not part of this package: pass
``positions`` (an (nant, 2) array of E, N coordinates in metres) or ``positions_file`` (a text table
``numpy.loadtxt`` reads, e.g. the reference's gmrtpositions.dat)."""
import numpy as np

from . import config, telescope


def gaussian_beam(angpos, pointing, sigma):
    """exp(-sin^2(angle from pointing) / (4 sigma^2)) (gmrt.py:97-107, focalplane.py:35-41)."""
    x2 = (1.0 - telescope.sph_dot(angpos, pointing) ** 2) / (4.0 * sigma**2)
    return np.exp(-x2)


class GmrtArray(telescope.TransitTelescope):
    fwhm = config.Property(proptype=float, default=3.1)           # degrees at 150 MHz
    pointing = config.Property(proptype=float, default=0.0)       # declination of the beam centre, degrees
    dish_width = config.Property(proptype=float, default=45.0)
    positions_file = config.Property(proptype=str, default=None)

    freq_lower = config.Property(proptype=float, default=139.33)
    freq_upper = config.Property(proptype=float, default=156.00)
    num_freq = config.Property(proptype=int, default=64)
    tsys_flat = config.Property(proptype=float, default=582.0, key="tsys")
    minlength = config.Property(proptype=float, default=0.0)
    maxlength = config.Property(proptype=float, default=600.0)

    def __init__(self, pointing=0.0, positions=None, positions_file=None):
        super(GmrtArray, self).__init__(latitude=19.09, longitude=74.05)
        self.pointing = pointing
        self._positions = None if positions is None else np.array(positions, dtype=np.float64).reshape(-1, 2)
        if positions_file is not None:
            self.positions_file = positions_file

    @property
    def u_width(self):
        return self.dish_width

    @property
    def v_width(self):
        return self.dish_width

    def beam(self, feed, freq):
        sigma = np.radians(self.fwhm) / (8.0 * np.log(2.0)) ** 0.5 / (self.frequencies[freq] / 150.0)
        pointing = np.array([np.pi / 2.0 - np.radians(self.pointing), self.zenith[1]])
        return gaussian_beam(self._angpos, pointing, sigma)

    beamx = beam
    beamy = beam

    @property
    def _single_feedpositions(self):
        if self._positions is None:
            if not self.positions_file:
                raise ValueError("GmrtArray needs `positions` or `positions_file` (the antenna table is not bundled)")
            self._positions = np.loadtxt(self.positions_file).reshape(-1, 2)
        return self._positions


class GmrtUnpolarised(GmrtArray, telescope.SimpleUnpolarisedTelescope):
    """Unpolarised GMRT class (gmrt.py:143-146)."""
