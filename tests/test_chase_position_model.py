"""The index-level model of the bulge chase by band position (scratch/proto_chase_pos.py): the numpy restatement of WHAT
`sb_chase_pos_kernel` (csrc/dm_sbr_impl.h) moves between positions — slot coordinates, the slide of the windows, the
reflector / beta / first-row / D-column / D-corner packets, mailbox depths — against the plain sweep-by-sweep band
reduction (LAPACK's zhbtrd scheme, the arithmetic behind scipy.linalg.eigh in kltransform.py:89).  Runs on the CPU: the kernel
itself is checked on the GPU against numpy's eigenvalues (tests/test_gpu_primitives.py, scratch/twostage_check.py)."""
import os
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "scratch"))
import proto_chase_pos as pcp  # noqa: E402


def _band(n, seed):
    rng = np.random.default_rng(seed)
    B = rng.standard_normal((n, n)) + 1j * rng.standard_normal((n, n))
    B = B + B.conj().T
    A = np.zeros((n, n), complex)
    for k in range(-min(pcp.SB, n - 1), min(pcp.SB, n - 1) + 1):
        A += np.diag(np.diag(B, k), k)
    return A


@pytest.mark.parametrize("n", [2, 3, 33, 34, 35, 64, 65, 66, 97, 129])
def test_by_position_equals_sweep_by_sweep(n):
    """d, e, every reflector and tau equal to rounding; the tridiagonal has the band's spectrum.  The sizes cover one position
    only (n <= 33), a second position with a one-row last block (34: no reflector there), exact multiples of the bandwidth and
    the sizes around them (the `rightOn` / `nr == 1` edges of the kernel)."""
    A = _band(n, 100 + n)
    d0, e0, r0 = pcp.chase_reference(A)
    d1, e1, r1 = pcp.chase_by_position(A)
    T = np.diag(d0) + np.diag(e0[: n - 1], 1) + np.diag(e0[: n - 1], -1)
    assert np.abs(np.linalg.eigvalsh(T) - np.linalg.eigvalsh(A)).max() < 1e-12 * n * np.abs(A).max()
    assert set(r0) == set(r1)
    scale = np.abs(A).max()
    assert np.abs(d0 - d1).max() < 1e-11 * scale and np.abs(e0 - e1).max() < 1e-11 * scale
    for k in r0:
        assert np.abs(r0[k][0] - r1[k][0]).max() < 1e-11 and abs(r0[k][1] - r1[k][1]) < 1e-11, k
