"""Host-side wall time of the phases of dm_jacobi_rows during the configs[1] SVD stage (DM_DEBUG=1 DM_DEBUG_NOSYNC=1:
marks without stream syncs = what the host spends issuing each phase)."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import tempfile
import bench
from driftscan_amd import btgen, device
import torch
ctx = device.get_context(workspace_bytes=24 << 30)
with tempfile.TemporaryDirectory() as tmp:
    tel, bt, kl = bench.build_objects(tmp)
    beam = btgen.beam_m_all(tel, ctx=ctx)
    for i in range(3):
        res = bt.svd_device(beam)
        ctx.sync()
    os.environ["DM_DEBUG"] = "1"; os.environ["DM_DEBUG_NOSYNC"] = "1"
    torch.cuda.synchronize(); t0 = time.perf_counter()
    res = bt.svd_device(beam)
    ctx.sync(); torch.cuda.synchronize()
    print("svd_device wall %.2f ms" % (1e3 * (time.perf_counter() - t0)), file=sys.stderr)
