"""driftscan_amd — MI355X (gfx950) implementation of driftscan's per-m hot path.

Host side: Python mirroring the reference's operator API (`ProductManager`,
`BeamTransfer`, `KLTransform`, `DoubleKL`, cylinder telescopes).  Compute side:
hand-written HIP kernels in ``csrc/`` behind the C ABI of ``include/driftmi.h``,
loaded through ctypes by ``driftscan_amd._lib``.  There is no CPU fallback: any
compute call without the built library and a GPU raises.
"""

__version__ = "0.1.0"
