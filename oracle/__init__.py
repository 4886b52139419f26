"""oracle/ — CPU restatement of driftscan's per-m hot path.  TEST INFRASTRUCTURE.

Only ``tests/``, ``__graft_entry__.smoke()`` and ``bench.py``'s ``cpu_baseline``
leg may import this package, and only as the checker / the timed CPU baseline.
The product (``driftscan_amd/``) never imports it and has no CPU fallback.

Pinning status:
  * svdchain / kl / geometry / pixel kernels: pinned against the *unmodified*
    reference imported in the build container (``oracle/gen_golden.py`` →
    ``tests/golden/*.npz``) and, for the pixel kernels, against the reference's
    own compiled Cython extension (``oracle/_ref``).
  * sht (HEALPix quadrature) and the analytic sky covariances: PARITY UNPINNED —
    the reference delegates these to healpy/libsharp and cora, which are neither
    vendored nor installed, and whose golden tarball is network-only
    (SURVEY.md §8c).  They are restated from the published algorithms and
    checked against brute-force spherical-harmonic sums instead.
"""
