"""Sky covariance models C_l(nu, nu') as [pol, pol, l, freq, freq] arrays.

The reference builds these from ``cora`` (drift/core/skymodel.py:20-68: 21 cm signal,
galactic synchrotron, point sources).  cora is neither vendored by the reference nor
installed here, and its constants cannot be verified in this environment, so this
module provides the *committed analytic model* named in SURVEY.md §8(d) with the
same call signatures and array layout:

  signal      C_l = A_s / (l + 1) * exp(-(dnu / nu_c)^2 / 2)                       on (T,T)
  foreground  C_l = A_f ((l + 1)/100)^-alpha (nu nu'/nu_0^2)^-beta exp(-ln^2(nu/nu')/(2 zeta^2))
              on (T,T), and pol_frac times the same with a shorter zeta on (E,E), (B,B)

(the functional form of cora's gaussianfg).  The per-m operators take the arrays as
inputs, so any other model — including cora's when available — can be injected
through ``KLTransform._cvsg`` / ``_cvfg`` exactly as with the reference.
"""
import numpy as np

# amplitudes in K^2; foregrounds dominate the signal by ~1e4 in power at l ~ 100, and the noise
# covariance of the benchmark telescopes stays conditioned like 1e9-1e10 (eps * cond ~ 1e-6:
# see tests/parity_util.pencil_tol for what that means for parity)
SIGNAL_AMP = 1e-11
SIGNAL_NUC = 2.0  # MHz
FG_AMP = 1e-9
FG_ALPHA = 2.4
FG_BETA = 2.8
FG_NU0 = 408.0
FG_ZETA = 4.0
POL_FRAC = 0.05
POL_ZETA = 0.5


def im21cm_model(lmax, frequencies, npol, cr=None, temponly=False):
    nu = np.asarray(frequencies, dtype=np.float64)
    ell = np.arange(lmax + 1, dtype=np.float64)
    dnu = nu[:, None] - nu[None, :]
    cv_t = SIGNAL_AMP * (1.0 / (ell + 1.0))[:, None, None] * np.exp(-0.5 * (dnu / SIGNAL_NUC) ** 2)[None]
    if temponly:
        return cv_t
    cv = np.zeros((npol, npol, lmax + 1, nu.size, nu.size))
    cv[0, 0] = cv_t
    return cv


def foreground_model(lmax, frequencies, npol, pol_frac=1.0, pol_length=None):
    nu = np.asarray(frequencies, dtype=np.float64)
    ell = np.arange(lmax + 1, dtype=np.float64)
    lognu = np.log(nu[:, None] / nu[None, :])
    spec = (nu[:, None] * nu[None, :] / FG_NU0**2) ** -FG_BETA
    amp = FG_AMP * ((ell + 1.0) / 100.0) ** -FG_ALPHA
    cv = np.zeros((npol, npol, lmax + 1, nu.size, nu.size))
    cv[0, 0] = amp[:, None, None] * (spec * np.exp(-0.5 * lognu**2 / FG_ZETA**2))[None]
    if npol >= 3:
        zeta = POL_ZETA if pol_length is None else pol_length
        polcv = POL_FRAC * amp[:, None, None] * (spec * np.exp(-0.5 * lognu**2 / zeta**2))[None]
        cv[1, 1] = pol_frac * polcv
        cv[2, 2] = pol_frac * polcv
    return cv
