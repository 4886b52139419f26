"""BASELINE configs[4], the KL stage on a REAL block: the m = 300 block of the CHIME-like telescope goes through BT-gen, the
SVD chain of all 256 frequencies and KLTransform — one generalised eigenproblem of order ~32 600 in a ~140 GB arena
(`dm_eigh_gen`: Cholesky, two triangular solves, two-stage tridiagonalisation, divide & conquer with merge nodes far beyond
the LDS, back-transformation of the kept modes).  Runs last and alone (the file name sorts after every other GPU test):
the eigensolver needs the card to itself.  Checked: the size-independent properties of the products (kltransform.py:310-355):
E N E^H = I, E S E^H = diag(lambda) on a sample of the kept modes, at the conditioning bound of the pencil."""
import os
import sys
import time

import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


def test_config5_real_block_through_kl():
    import bench

    def log(*a):
        print(time.strftime("%H:%M:%S"), *a, flush=True)

    rec = bench.measure_configs4_block(300, checks=True, log=log)
    log("configs[4] m = 300: BT-gen %.1f s, SVD chain %.1f s, KL %.1f s (n = %d, %d modes kept), HBM peak %.0f GB"
        % (rec["btgen_s"], rec["svd_s"], rec["kl_s"], rec["ndof"], rec["kl_nkept"], rec["hbm_peak_gb"]))
    assert rec["nbase"] == 1776 and rec["ndof"] > 30000
    assert rec["check_ut_orth"] < 1e-12 and rec["check_beam_pinv"] < 1e-7
    assert rec["kl_nkept"] > 0 and rec["kl_add_const"] == 0.0
    # m = 300: cond(N) ~ 1e5 (round 1 measured 7.3e-12 / 1.6e-15 on this pencil)
    assert rec["check_ENE"] < 1e-9 and rec["check_ESE_offdiag"] < 1e-9 and rec["check_ESE_diag"] < 1e-9
