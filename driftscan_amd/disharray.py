"""Interferometric arrays of dishes on a regular grid (drift/telescope/disharray.py, and the complete class of
examples/disharray/simplearray.py): uniformly illuminated circular apertures, feeds on a gridu x gridv lattice.

The beams go through the host plug-in interface of ``telescope.TransitTelescope`` (``beam`` / ``beamx`` /
``beamy`` on ``self._angpos``): one map per (frequency, beam class) is uploaded and the per-baseline work —
fringes, products, ring DFT, Legendre transform — runs on the device as for the cylinders.
"""
import numpy as np
from scipy.special import jn

from . import config, telescope


def jinc(x):
    """J1(x)/x through the recurrence J0 + J2 = 2 J1 / x (no division at x = 0)."""
    return 0.5 * (jn(0, x) + jn(2, x))


def beam_circular(angpos, zenith, uv_diameter):
    """Field pattern 2 jinc(pi D sin(angle from zenith)) of a uniformly illuminated circular dish
    `uv_diameter` wavelengths across (disharray.py:13-33)."""
    x = (1.0 - telescope.sph_dot(angpos, zenith) ** 2) ** 0.5 * np.pi * uv_diameter
    return 2.0 * jinc(x)


class DishArray(telescope.TransitTelescope):
    """Feed grid and aperture of a dish array; combine with a Simple(Un)polarisedTelescope."""

    dish_width = config.Property(proptype=float, default=3.5)
    gridu = config.Property(proptype=int, default=4)
    gridv = config.Property(proptype=int, default=4)

    @property
    def u_width(self):
        return self.dish_width

    @property
    def v_width(self):
        return self.dish_width

    def _amplitude(self, freq):
        return beam_circular(self._angpos, self.zenith, self.dish_width / self.wavelengths[freq])

    @property
    def _single_feedpositions(self):
        iu, iv = np.meshgrid(np.arange(self.gridu), np.arange(self.gridv), indexing="ij")
        return np.stack([iu.ravel(), iv.ravel()], axis=1) * float(self.dish_width)


class UnpolarisedDishArray(DishArray, telescope.SimpleUnpolarisedTelescope):
    """disharray.py:36-157 completed with the unpolarised mixin."""

    freq_lower = config.Property(proptype=float, default=1000.0)
    freq_upper = config.Property(proptype=float, default=1200.0)
    num_freq = config.Property(proptype=int, default=100)

    def beam(self, feed, freq):
        return self._amplitude(freq)


class PolarisedDishArray(DishArray, telescope.SimplePolarisedTelescope):
    """examples/disharray/simplearray.py:36-107: X dipoles along phihat (E-W), Y dipoles along thetahat (N-S)."""

    freq_lower = config.Property(proptype=float, default=100.0)
    freq_upper = config.Property(proptype=float, default=150.0)
    num_freq = config.Property(proptype=int, default=5)

    def beamx(self, feed, freq):
        return self._amplitude(freq)[:, np.newaxis] * np.array([0.0, 1.0])

    def beamy(self, feed, freq):
        return self._amplitude(freq)[:, np.newaxis] * np.array([1.0, 0.0])
