"""First (cold) against second (warm) call of the SVD chain on the same batch in one process — and the same with the
output blocks touched beforehand (torch.empty + del: the caching allocator keeps them)."""
import os, sys, time, tempfile
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np, torch
import bench
from driftscan_amd import beamtransfer, btgen, cylinder, device
ctx = device.get_context(workspace_bytes=100 << 30)
tel = cylinder.PolarisedCylinderTelescope.from_config(dict(bench.CFG3))
bt = beamtransfer.BeamTransfer(tempfile.mkdtemp(), telescope=tel)
n = 9
beam = btgen.beam_m_all(tel, ctx=ctx, max_bytes=48 << 30, m_range=(0, n - 1))
ctx.sync()
if os.environ.get("PRETOUCH"):
    F, T, P, L, K = tel.nfreq, bt.ntel, tel.num_pol_sky, tel.lmax + 1, bt.svd_len
    t0 = time.perf_counter()
    held = [torch.empty(s, dtype=torch.complex128, device="cuda") for s in ((n, F, K, P, L), (n, F, P, L, K), (n, F, K, T))]
    for h in held: h.zero_()
    torch.cuda.synchronize()
    print("pretouch (alloc + zero) %.3f s" % (time.perf_counter() - t0))
    del held
for rep in range(3):
    ctx.prof_reset(2)
    torch.cuda.synchronize(); t0 = time.perf_counter()
    res = bt.svd_device(beam, ms=list(range(n)))
    ctx.sync(); torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    ks = sum(v["ms"] for v in ctx.prof_report().values()) * 1e-3
    print("call %d: wall %.3f s, kernel classes %.3f s, gap %.3f s" % (rep, dt, ks, dt - ks), flush=True)
    del res
