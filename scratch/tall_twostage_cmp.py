"""Singular values of the tall SVD chain with the Gram eigenproblems on the one-stage (default) and on the two-stage
reduction (DM_SVD_TALL_TWOSTAGE=1): files written by svd_phase_probe.py (PROBE_SAVE), compared here."""
import sys, numpy as np
a, b = np.load(sys.argv[1]), np.load(sys.argv[2])
sa, sb = a["sv"], b["sv"]
print("nmodes equal:", np.array_equal(a["nmodes"], b["nmodes"]), "| max |dsigma| / sigma_max:", np.abs(sa - sb).max() / np.abs(sa).max(),
      "| finite:", np.isfinite(sa).all(), np.isfinite(sb).all())
