"""cora.util.hputil wraps healpy; neither is available here."""
import numpy as np


def nside_for_lmax(lmax, accuracy_boost=1):
    return int(2 ** (accuracy_boost + np.ceil(np.log((lmax + 1) / 3.0) / np.log(2.0))))


def _na(*a, **k):
    raise NotImplementedError("healpy/cora SHT is not available in this container")


ang_positions = sphtrans_complex = sphtrans_complex_pol = _na
sphtrans_sky = sphtrans_inv_sky = sphtrans_real = _na
