"""Gauge-invariant comparison helpers shared by the oracle and GPU parity tests.

Singular / eigen *vectors* are only defined up to a phase (and a rotation inside
degenerate subspaces); the reference's own tests skip them for that reason
(tests/test_functional.py:212-235).  We therefore compare projectors and
reconstructed matrices, and spectra against an absolute scale.
"""
import numpy as np


def relerr(a, b, scale=None):
    a = np.asarray(a)
    b = np.asarray(b)
    if scale is None:
        scale = max(np.abs(b).max() if b.size else 0.0, 1e-300)
    return (np.abs(a - b).max() / scale) if a.size else 0.0


def assert_spectrum(a, b, rtol=1e-10, what=""):
    a = np.asarray(a)
    b = np.asarray(b)
    assert a.shape == b.shape, "%s shape %s vs %s" % (what, a.shape, b.shape)
    if a.size == 0:
        return
    scale = np.abs(b).max()
    err = np.abs(a - b).max()
    assert err <= rtol * max(scale, 1e-300), "%s: max abs err %.3e vs scale %.3e" % (what, err, scale)


def rowspace_projector(rows):
    """Projector onto the span of the rows of a (k, n) matrix with orthonormal rows."""
    return rows.T @ rows.conj()


def assert_same_rowspace(a, b, tol=1e-9, what=""):
    assert a.shape == b.shape, "%s shape %s vs %s" % (what, a.shape, b.shape)
    if a.size == 0:
        return
    pa, pb = rowspace_projector(a), rowspace_projector(b)
    err = np.abs(pa - pb).max()
    assert err <= tol, "%s: projector mismatch %.3e" % (what, err)


def pencil_tol(B, base=1e-10, factor=50.0):
    """Eigenvalue tolerance (relative to the largest eigenvalue) for the pencil (A, B).

    The Cholesky reduction A -> L^-1 A L^-H used by LAPACK's zhegvd (the reference,
    kltransform.py:89) and by our solver is only conditionally stable: the computed
    spectrum carries an error of order eps * cond(B) * |lambda_max| whatever the
    implementation.  For cond(B) <~ 1e5 this is below the 1e-10 target; for the
    foreground-dominated matrices of a real KL run (cond(B) ~ 1e12+) no two
    implementations - not even two LAPACK builds - agree to 1e-10, and the bound
    below is what parity can mean.
    """
    w = np.linalg.eigvalsh(0.5 * (B + B.conj().T))
    cond = abs(w[-1]) / max(abs(w[0]), 1e-300)
    return max(base, factor * 2.220446049250313e-16 * cond)


def pencil_sensitivity(A, B, nrep=4, seed=0):
    """Empirical conditioning of the generalised eigenvalues: how far LAPACK's own
    answer moves (relative to |lambda_max|) when A and B are perturbed elementwise
    at the level of one unit roundoff.  An implementation-independent error bar."""
    import scipy.linalg as la

    rng = np.random.default_rng(seed)
    ref = la.eigh(A, B, eigvals_only=True)
    scale = np.abs(ref).max()
    worst = 0.0
    for _ in range(nrep):
        def pert(M):
            d = 1.0 + 2.2e-16 * rng.standard_normal(M.shape)
            Mp = M * d
            return 0.5 * (Mp + Mp.conj().T)
        ev = la.eigh(pert(A), pert(B), eigvals_only=True)
        worst = max(worst, np.abs(ev - ref).max() / scale)
    return worst
