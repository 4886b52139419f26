"""CPU restatement (numpy/scipy) of driftscan's per-m covariance projection,
Karhunen-Loeve transform and DoubleKL foreground filter.

TEST INFRASTRUCTURE: the checker for the HIP path and the timed ``cpu_baseline``
of bench.py.  Never imported by the product.

Pinned: ``tests/golden/kl_*.npz`` were produced by the unmodified reference
(``oracle/gen_golden.py``); ``tests/test_oracle_golden.py`` checks this file against
them.

Follows (paths relative to the reference tree):
  * ``project_matrix_sky_to_svd``                 drift/core/beamtransfer.py:1135-1188
  * ``project_matrix_diagonal_telescope_to_svd``  drift/core/beamtransfer.py:1190-1231
  * ``sn_covariance``                             drift/core/kltransform.py:258-308
  * ``eigh_gen``                                  drift/core/kltransform.py:55-121
  * ``kl_transform_m`` / ``threshold_cut``        drift/core/kltransform.py:310-355, 385-398
  * ``doublekl_transform_m``                      drift/core/doublekl.py:30-87
"""
import re

import numpy as np
import scipy.linalg as la

from .svdchain import svd_num


def project_matrix_sky_to_svd(beam_svd, svnum, svbounds, mat, temponly=False):
    """Sky covariance ``mat (P,P,L,F,F)`` (real) projected into the SVD basis.

    ``out[b_f:b_f+n_f, b_f':b_f'+n_f'] += (B[f,:n_f,pi,:] * C[pi,pj,:,f,f']) @
    B[f',:n_f',pj,:]^H`` over all pol pairs (beamtransfer.py:1168-1186).
    """
    F, K, P, L = beam_svd.shape
    npol = 1 if temponly else P
    ndof = int(svbounds[-1])
    out = np.zeros((ndof, ndof), dtype=np.complex128)
    freqs = [f for f in range(F) if svnum[f] > 0]
    for pi in range(npol):
        for pj in range(npol):
            for fi in freqs:
                bi = beam_svd[fi, : svnum[fi], pi, :]
                for fj in freqs:
                    bj = beam_svd[fj, : svnum[fj], pj, :]
                    out[svbounds[fi] : svbounds[fi + 1], svbounds[fj] : svbounds[fj + 1]] += (
                        bi * mat[pi, pj, :, fi, fj]
                    ) @ bj.T.conj()
    return out


def project_matrix_diagonal_telescope_to_svd(beam_ut, svnum, svbounds, dmat):
    """Diagonal telescope-basis matrix ``dmat (F,T)`` into the SVD basis:
    block-diagonal ``(U_f * d_f) U_f^H`` (beamtransfer.py:1217-1227)."""
    F = beam_ut.shape[0]
    ndof = int(svbounds[-1])
    out = np.zeros((ndof, ndof), dtype=np.complex128)
    for fi in range(F):
        if svnum[fi] == 0:
            continue
        u = beam_ut[fi, : svnum[fi], :]
        out[svbounds[fi] : svbounds[fi + 1], svbounds[fi] : svbounds[fi + 1]] = (u * dmat[fi]) @ u.T.conj()
    return out


def sn_covariance(
    beam_svd,
    beam_ut,
    singularvalues,
    cv_sg,
    cv_fg,
    npower,
    svcut=1e-6,
    regulariser=1e-14,
    use_thermal=True,
    use_foregrounds=True,
    tsys_flat=None,
):
    """Signal and noise covariances in the SVD basis (kltransform.py:258-308).

    ``npower`` is ``telescope.noisepower(bl, f)`` on the duplicated baseline axis,
    shape (F, T).  When ``use_thermal`` is False the noise is scaled by
    ``(1e-3 / tsys_flat)**2`` (kltransform.py:294-296).
    """
    if not (use_foregrounds or use_thermal):
        raise Exception("Either `use_thermal` or `use_foregrounds`, or both must be True.")
    svnum, svbounds = svd_num(singularvalues, svcut)
    cvb_s = project_matrix_sky_to_svd(beam_svd, svnum, svbounds, cv_sg)
    if use_foregrounds:
        cvb_n = project_matrix_sky_to_svd(beam_svd, svnum, svbounds, cv_fg)
    else:
        cvb_n = np.zeros_like(cvb_s)
    # regulariser: numpy's max of a complex array is lexicographic in (re, im)
    cvb_n[np.diag_indices_from(cvb_n)] += regulariser * cvb_n.max()
    nc = 1.0
    if not use_thermal:
        nc = (1e-3 / tsys_flat) ** 2
    cvb_n += project_matrix_diagonal_telescope_to_svd(beam_ut, svnum, svbounds, nc * npower)
    return cvb_s, cvb_n


def eigh_gen(A, B):
    """Generalised Hermitian-definite eigenproblem ``A v = lambda B v`` with the
    reference's rescue for a numerically non-positive-definite ``B``
    (kltransform.py:55-121).  Returns (evals ascending, evecs as columns, add_const).
    """
    add_const = 0.0
    if (A == 0).all():
        return np.zeros(A.shape[0]), np.identity(A.shape[0], dtype=A.dtype), add_const
    A = A.copy()
    B = B.copy()
    try:
        evals, evecs = la.eigh(A, B)
    except la.LinAlgError as e:
        mo = re.search(r"order (\d+)", e.args[0])
        if mo is None:
            raise
        if int(mo.group(1)) < (A.shape[0] + 1):
            evb = la.eigvalsh(B)
            add_const = 1e-15 * evb[-1] - 2.0 * evb[0] + 1e-60
            B[np.diag_indices(B.shape[0])] += add_const
            evals, evecs = la.eigh(A, B)
        else:
            evals, evecs = la.eigh(A, B, driver="gv")
    return evals, evecs, add_const


def kl_transform_m(cvb_s, cvb_n):
    """KLTransform._transform_m after the covariances are built
    (kltransform.py:339-345): rows of the returned ``evecs`` are the modes."""
    n = cvb_s.shape[0]
    if n == 0:
        return np.array([]), np.array([[]]), 0.0
    evals, evecs, ac = eigh_gen(cvb_s, cvb_n)
    return evals, evecs.T.conj(), ac


def threshold_cut(evals, evecs, threshold, ndof=None):
    """transform_save's padding and S/N cut (kltransform.py:385-398).

    Returns (evals_full padded at the low end to ``ndof``, kept evals, kept evecs).
    """
    ndof = evals.size if ndof is None else ndof
    evals_full = np.zeros(ndof, dtype=np.float64)
    if evals.size != 0:
        evals_full[-evals.size :] = evals
    i_ev = int(np.searchsorted(evals, threshold))
    return evals_full, evals[i_ev:], evecs[i_ev:]


def inv_gen(A):
    """Inverse with the pseudo-inverse as fallback (kltransform.py:124-143)."""
    try:
        return la.inv(A)
    except la.LinAlgError:
        return la.pinv(A)


def kl_inverse(evecs_rows):
    """The `inv` of KLTransform._transform_m: inv_gen(evecs).T (kltransform.py:346-347)."""
    return inv_gen(evecs_rows).T


def doublekl_transform_m(sn_cov, foreground_threshold=100.0, inverse=False):
    """DoubleKL._transform_m (doublekl.py:30-87).

    ``sn_cov(use_thermal)`` must return the (S, N) pair for the given flag —
    the reference calls ``sn_covariance`` twice, first with ``use_thermal=False``
    and then with ``True``.
    Returns (evals, evecs [rows are modes], f_evals, ac) — `ac` is the STAGE-1 shift, the one the
    reference stores in `evextra` (doublekl.py:58) — and, with ``inverse``, `inv` as a fifth item
    (doublekl.py:63-67, :83-85: ``inv_gen(evecs2) @ inv_gen(E1).T[ind]``).
    """
    cs, cn = sn_cov(False)
    if cs.shape[0] == 0:
        out = (np.array([]), np.array([[]]), np.array([]), 0.0)
        return out + (np.array([[]]),) if inverse else out
    evals, evecs2, ac = eigh_gen(cs, cn)
    evecs = evecs2.T.conj()
    f_evals = evals.copy()
    ac1 = ac
    ind = np.where(evals > foreground_threshold)
    inv = inv_gen(evecs).T[ind] if inverse else None
    evals = evals[ind]
    evecs = evecs[ind]
    if evals.size > 0:
        cs, cn = sn_cov(True)
        cs = evecs @ (cs @ evecs.T.conj())
        cn = evecs @ (cn @ evecs.T.conj())
        evals, evecs2, ac = eigh_gen(cs, cn)
        evecs = evecs2.T.conj() @ evecs
        if inverse:
            inv = inv_gen(evecs2) @ inv
    out = (evals, evecs, f_evals, ac1)
    return out + (inv,) if inverse else out
