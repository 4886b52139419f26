"""CPU restatement of the exact per-m Fisher matrix of the band powers
(PSExact, drift/core/psestimation.py:672-815).  TEST INFRASTRUCTURE — never imported by the
product.  Pinned against tests/golden/psfisher.npz (outputs of the unmodified reference's
`PSExact._work_fisher_bias_m` with injected band C_l arrays)."""
import numpy as np

from . import kl as okl


def fisher_m(beam_svd, svnum, svbounds, evals, evecs, clarray):
    """fisher[a, b] = sum_ij C_a[i, j] C_b[j, i] / ((lam_i + 1)(lam_j + 1)),
    C_a = E (B C_l^a B^H) E^H  (psestimation.py:672-699 makeproj, :775-815).

    beam_svd (F, K, P, L); evals (n,), evecs (n, ndof) = KL modes above the threshold;
    clarray (nbands, L, F, F).  Returns (fisher complex (nbands, nbands), bias zeros)."""
    nb = clarray.shape[0]
    fisher = np.zeros((nb, nb), dtype=np.complex128)
    bias = np.zeros(nb, dtype=np.complex128)
    proj = []
    for a in range(nb):
        cl = clarray[a].reshape((1, 1) + clarray[a].shape)
        svdmat = okl.project_matrix_sky_to_svd(beam_svd, svnum, svbounds, cl, temponly=True)
        proj.append(evecs @ svdmat @ evecs.T.conj())  # kltransform.py:794-818 project_matrix_svd_to_kl
    ci = 1.0 / (evals + 1.0) ** 0.5
    ci = np.outer(ci, ci)
    for ia in range(nb):
        c_a = proj[ia]
        fisher[ia, ia] = np.sum(c_a * c_a.T * ci**2)
        for ib in range(ia):
            fisher[ia, ib] = np.sum(c_a * proj[ib].T * ci**2)
            fisher[ib, ia] = np.conj(fisher[ia, ib])
    return fisher, bias


def band_clarray(frequencies, L, l_edges, nu_c=0.08, amp=1e-11):
    """Analytic stand-in for cora's band angular power spectra (cora is not available; the real
    bands are an INPUT of the estimator): band a is the signal model
    A (l+1)^-1 exp(-(dnu/nu)^2 / (2 nu_c^2)) restricted to l_edges[a] <= l < l_edges[a+1]."""
    f = np.asarray(frequencies, dtype=np.float64)
    dn = (f[:, None] - f[None, :]) / f.mean()
    corr = np.exp(-0.5 * (dn / nu_c) ** 2)
    ell = np.arange(L)
    out = np.zeros((len(l_edges) - 1, L, f.size, f.size))
    for a in range(len(l_edges) - 1):
        sel = (ell >= l_edges[a]) & (ell < l_edges[a + 1])
        out[a, sel] = (amp / (ell[sel] + 1.0))[:, None, None] * corr[None]
    return out
