#!/usr/bin/env python3
"""Host time inside a BT-gen range call of the configs[2] job (cProfile of the driver thread + wall / kernel seconds)."""
import cProfile, io, os, pstats, sys, tempfile, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench
import torch
from driftscan_amd import btgen, cylinder, device
a, b = int(sys.argv[1]), int(sys.argv[2])
ctx = device.get_context(workspace_bytes=100 << 30)
tel = cylinder.PolarisedCylinderTelescope.from_config(dict(bench.CFG3))
for rep in range(2):
    ctx.prof_reset(2); torch.cuda.synchronize()
    pr = cProfile.Profile(); t0 = time.perf_counter(); pr.enable()
    beam = btgen.beam_m_all(tel, ctx=ctx, max_bytes=48 << 30, m_range=(a, b))
    t1 = time.perf_counter()
    ctx.sync(); torch.cuda.synchronize(); pr.disable(); t2 = time.perf_counter()
    rep_ = ctx.prof_report(); ks = sum(v["ms"] for v in rep_.values()) * 1e-3
    print("rep %d: m = %d..%d  call returned after %.2f s, device idle after %.2f s, kernels %.2f s" % (rep, a, b, t1 - t0, t2 - t0, ks), flush=True)
    s = io.StringIO(); pstats.Stats(pr, stream=s).sort_stats("cumulative").print_stats(18); print(s.getvalue()[:3500], flush=True)
    del beam
