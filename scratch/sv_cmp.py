"""Singular values of two PROBE_SAVE files: per chain max |d sigma| / sigma_max, and the largest relative change among the
modes above 1e-6 sigma_max."""
import sys, numpy as np
a, b = np.load(sys.argv[1]), np.load(sys.argv[2])
sa, sb = a["sv"], b["sv"]
sa = sa.reshape(-1, sa.shape[-1]); sb = sb.reshape(-1, sb.shape[-1])
mx = sa.max(axis=1, keepdims=True)
d = np.abs(sa - sb) / mx
keep = sa > 1e-6 * mx
rel = np.where(keep, np.abs(sa - sb) / np.where(sa > 0, sa, 1), 0)
print("nmodes equal:", np.array_equal(a["nmodes"], b["nmodes"]), "| max |dsigma|/sigma_max %.2e (median chain %.2e) | max relative change above 1e-6 sigma_max: %.2e"
      % (d.max(), np.median(d.max(axis=1)), rel.max()))
