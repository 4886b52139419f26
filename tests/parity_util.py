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
