"""Cylinder variants with irregular feed layouts or perturbed beams (drift/telescope/exotic_cylinder.py)."""
import numpy as np

from . import config, cylinder


class RandomCylinder(cylinder.UnpolarisedCylinderTelescope):
    """Feed positions jittered N-S by pos_sigma x feed_spacing (Gaussian, seeded by the cylinder index, sorted)."""

    pos_sigma = config.Property(proptype=float, default=0.5)

    def feed_positions_cylinder(self, cylinder_index):
        pos = super(RandomCylinder, self).feed_positions_cylinder(cylinder_index)
        # the reference draws from the legacy global generator seeded with the cylinder index
        # (exotic_cylinder.py:13-29); RandomState(seed) yields the same stream without touching global state
        noise = np.random.RandomState(cylinder_index).standard_normal(pos.shape[0])
        pos[:, 1] = np.sort(pos[:, 1] + self.pos_sigma * self.feed_spacing * noise)
        return pos


class GradientCylinder(cylinder.UnpolarisedCylinderTelescope):
    """Feed spacing growing linearly along the cylinder: y_i = a i + b i^2 / 2 (exotic_cylinder.py:32-55)."""

    min_spacing = config.Property(proptype=float, default=-1.0)
    max_spacing = config.Property(proptype=float, default=20.0)

    def feed_positions_cylinder(self, cylinder_index):
        if cylinder_index >= self.num_cylinders or cylinder_index < 0:
            raise Exception("Cylinder index is invalid.")
        nf = self.num_feeds
        a = self.wavelengths[-1] / 2.0 if self.min_spacing < 0.0 else self.min_spacing
        b = 2.0 * (self.max_spacing - a * (nf - 1)) / (nf - 1) ** 2.0
        i = np.arange(nf)
        pos = np.empty([nf, 2], dtype=np.float64)
        pos[:, 0] = cylinder_index * self.cylinder_spacing
        pos[:, 1] = a * i + 0.5 * b * i**2
        return pos


class CylinderExtra(cylinder.UnpolarisedCylinderTelescope):
    """Extra feeds at given N-S positions in front of the regular ones of every cylinder."""

    extra_feeds = config.Property(proptype=np.array, default=[])

    def feed_positions_cylinder(self, cylinder_index):
        pos = super(CylinderExtra, self).feed_positions_cylinder(cylinder_index)
        extra = np.asarray(self.extra_feeds, dtype=np.float64).reshape(-1)
        head = np.stack([np.full(extra.size, cylinder_index * self.cylinder_spacing), extra], axis=1)
        return np.concatenate([head, pos])


class CylinderShift(CylinderExtra):
    """The reference's ``CylinderShift`` (exotic_cylinder.py:190-215) repeats ``CylinderExtra`` through a
    ``super(CylinderExtra, self)`` call that raises for its own instances; this is the working equivalent."""

    shift = config.Property(proptype=float, default=0.0)


class CylinderPerturbed(cylinder.PolarisedCylinderTelescope):
    """Every feed twice over per perturbation order: beam classes 0, 1 are the X, Y beams, classes 2, 3 their
    first derivatives with respect to the E-plane width (finite difference at +1 %), exotic_cylinder.py:75-187."""

    npert = 2

    def beam_spec(self, beamclass, freq_index):
        if int(beamclass) // 2 == 0:
            return cylinder.PolarisedCylinderTelescope.beam_spec(self, beamclass, freq_index)
        return None   # the derivative beams are differences of two device-evaluated patterns, formed on the host

    @property
    def beamclass(self):
        n = self._single_feedpositions.shape[0]
        return np.repeat(np.arange(2 * self.npert), n).astype(np.int64)

    @property
    def feedpositions(self):
        return np.concatenate([self._single_feedpositions] * (2 * self.npert))

    def _pert(self, feed, freq, kind):
        order = int(self.beamclass[feed]) // 2
        ew = (lambda fe: fe) if kind == 1 else (lambda fe: self.fwhm_h)
        ns = (lambda fe: self.fwhm_h) if kind == 1 else (lambda fe: fe)
        b0 = self._pattern_host(kind, ew(self.fwhm_e), ns(self.fwhm_e), freq)
        if order == 0:
            return b0
        if order == 1:
            b1 = self._pattern_host(kind, ew(self.fwhm_e * 1.01), ns(self.fwhm_e * 1.01), freq)
            return (b1 - b0) / (0.01 * self.fwhm_e)
        return None   # like the reference, orders above the first have no beam

    def beamx(self, feed, freq):
        return self._pert(feed, freq, 1)

    def beamy(self, feed, freq):
        return self._pert(feed, freq, 2)
