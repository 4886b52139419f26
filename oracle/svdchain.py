"""CPU restatement (numpy/scipy) of driftscan's per-(m, frequency) SVD compression.

TEST INFRASTRUCTURE: the checker for the HIP path and the timed ``cpu_baseline``
of bench.py.  Never imported by the product.

Pinned: ``tests/golden/svdchain_*.npz`` were produced by the unmodified reference
(``oracle/gen_golden.py``); ``tests/test_oracle_golden.py`` checks this file against
them.

Follows (paths relative to the reference tree):
  * ``matrix_image``      drift/core/beamtransfer.py:68-104
  * ``matrix_nullspace``  drift/core/beamtransfer.py:107-143
  * ``svd_m``             drift/core/beamtransfer.py:730-929 (_generate_svdfile_m)
  * ``svd_num``           drift/core/beamtransfer.py:1116-1129 (_svd_num)
"""
import numpy as np
import scipy.linalg as la


def matrix_image(A, rtol=1e-8):
    """Orthonormal basis of the column space of ``A`` and its singular values.

    Keeps the left singular vectors whose singular value is strictly greater
    than ``rtol * s[0]`` (beamtransfer.py:98).  A 0-row input gives empty
    outputs (beamtransfer.py:69-70).
    """
    if A.shape[0] == 0:
        return np.zeros((0, 0), dtype=A.dtype), np.zeros(0)
    u, s, _ = la.svd(A, full_matrices=False)
    cut = int((s > s[0] * rtol).sum())
    return u[:, :cut].copy(), s


def matrix_nullspace(A, rtol=1e-8):
    """Orthonormal basis of the left null space of ``A`` and its singular values.

    Full SVD; drops the first ``cut`` left vectors where ``cut`` counts singular
    values greater than *or equal to* ``rtol * s[0]`` (beamtransfer.py:137 — note
    ``>=`` here versus ``>`` in ``matrix_image``).
    """
    if A.shape[0] == 0:
        return np.zeros((0, 0), dtype=A.dtype), np.zeros(0)
    u, s, _ = la.svd(A, full_matrices=True)
    cut = int((s >= s[0] * rtol).sum())
    return u[:, cut:].copy(), s


def svd_len(ntel, lmax):
    """beamtransfer.py:1443-1445."""
    return min(lmax + 1, ntel)


def svd_m(beam_m, noisew, polsvcut=1e-4, skip_svd_inv=False):
    """SVD-compress one m-block.

    Parameters
    ----------
    beam_m : (F, 2, B, P, L) complex128
        The m-ordered beam transfer block as returned by ``BeamTransfer.beam_m``
        (zero for l < m).
    noisew : (F, B) float64
        ``noisepower(b, f) ** -0.5`` (beamtransfer.py:810-812); duplicated for the
        two m signs inside (``:813``).
    polsvcut : float
        Relative threshold for the polarisation null space (``:846``).

    Returns
    -------
    dict with ``beam_svd (F,K,P,L)``, ``invbeam_svd (F,P,L,K)`` (or None),
    ``beam_ut (F,K,T)``, ``singularvalues (F,K)`` and ``nmodes (F,)``; rows past
    ``nmodes[f]`` are zero exactly as in the reference's zero-initialised HDF5
    datasets.
    """
    F, two, B, P, L = beam_m.shape
    assert two == 2
    T = 2 * B
    K = svd_len(T, L - 1)

    beam_svd = np.zeros((F, K, P, L), dtype=np.complex128)
    invbeam_svd = None if skip_svd_inv else np.zeros((F, P, L, K), dtype=np.complex128)
    beam_ut = np.zeros((F, K, T), dtype=np.complex128)
    sing = np.zeros((F, K), dtype=np.float64)
    nmodes_f = np.zeros(F, dtype=np.int64)

    for fi in range(F):
        nw = np.concatenate([noisew[fi], noisew[fi]])
        bf = beam_m[fi].reshape(T, P, L) * nw[:, None, None]
        bfr = bf.reshape(T, P * L)

        if P == 1:
            bf2 = bfr
            ut2 = np.identity(T, dtype=np.complex128)
            s1 = None
        else:
            # SVD 1: coarse projection onto the sky-sensitive telescope modes
            u1, s1 = matrix_image(bfr, rtol=1e-10)
            ut1 = u1.T.conj()
            bf1 = ut1 @ bfr
            # SVD 2: left null space of the polarised columns
            bfp = bf1.reshape(bf1.shape[0], P, L)[:, 1:].reshape(bf1.shape[0], (P - 1) * L)
            u2, _ = matrix_nullspace(bfp, rtol=polsvcut)
            ut2 = u2.T.conj() @ ut1
            bf2 = ut2 @ bfr

        if bf2.shape[0] > 0 and (P == 1 or (s1 > 0.0).any()):
            # SVD 3: decompose what is left using the total-intensity columns only
            bft = bf2.reshape(-1, P, L)[:, 0]
            u3, s3 = matrix_image(bft, rtol=0.0)
            ut3 = u3.T.conj() @ ut2
            nmodes = ut3.shape[0]
            if nmodes == 0:
                continue
            beam = ut3 @ bfr
            beam_ut[fi, :nmodes] = ut3 * nw[None, :]
            beam_svd[fi, :nmodes] = beam.reshape(nmodes, P, L)
            if not skip_svd_inv:
                invbeam_svd[fi, :, :, :nmodes] = la.pinv(beam).reshape(P, L, nmodes)
            sing[fi, :nmodes] = s3[:nmodes]
            nmodes_f[fi] = nmodes

    return dict(
        beam_svd=beam_svd,
        invbeam_svd=invbeam_svd,
        beam_ut=beam_ut,
        singularvalues=sing,
        nmodes=nmodes_f,
    )


def svd_num(singularvalues, svcut=1e-6):
    """Kept-mode counts and block bounds (beamtransfer.py:1116-1129).

    The cut is relative to the maximum over *all* frequencies of this m.
    """
    sv = np.asarray(singularvalues)
    svnum = (sv > sv.max() * svcut).sum(axis=1)
    svbounds = np.cumsum(np.insert(svnum, 0, 0))
    return svnum, svbounds
