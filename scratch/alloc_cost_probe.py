"""What the host does between the last BT-gen kernel and the first SVD kernel of a configs[2] share: cold and warm
torch allocations of the SVD output blocks, the noise weights, the C_l table upload."""
import os, sys, time, tempfile
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np, torch
import bench
from driftscan_amd import beamtransfer, cylinder, device
tel = cylinder.PolarisedCylinderTelescope.from_config(dict(bench.CFG3))
bt = beamtransfer.BeamTransfer(tempfile.mkdtemp(), telescope=tel)
ctx = device.get_context(workspace_bytes=100 << 30)
def T(label, fn):
    torch.cuda.synchronize(); t0 = time.perf_counter(); r = fn(); torch.cuda.synchronize()
    print("%-40s %.3f s" % (label, time.perf_counter() - t0), flush=True); return r
F, Tt, P, L, K = tel.nfreq, bt.ntel, tel.num_pol_sky, tel.lmax + 1, bt.svd_len
nb = 10
big = T("beam_all 52 GB (cold)", lambda: torch.empty((29, F, 2, tel.nbase, P, L), dtype=torch.complex128, device="cuda"))
T("zero it", lambda: big.zero_())
shapes = [(nb, F, K, P, L), (nb, F, P, L, K), (nb, F, K, Tt)]
held = T("SVD outputs 29 GB (cold)", lambda: [torch.empty(s, dtype=torch.complex128, device="cuda") for s in shapes])
del held
held = T("SVD outputs (warm, from the cache)", lambda: [torch.empty(s, dtype=torch.complex128, device="cuda") for s in shapes])
T("noise weights to the device", lambda: bt._noisew_device())
mat = np.random.default_rng(0).standard_normal((4, 4, L, F, F))
T("C_l table upload + layout (270 MB)", lambda: beamtransfer.BeamTransfer._cl_device(mat))
T("pinned copy of 270 MB", lambda: torch.from_numpy(mat).pin_memory())
