#!/usr/bin/env python3
"""BT-gen of a narrow m-range of the configs[4] telescope (nside 512, four Stokes maps) on a few frequencies: seconds and
kernel classes with the belt by FFT (default) and by the matrix form (DM_BT_FFT=0, read once per process).
    python scratch/btgen_narrow_probe.py <nfreq> <m0> <count>"""
import json, os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
import torch
import bench
from driftscan_amd import btgen, cylinder, device
nf, m0, cnt = int(sys.argv[1]), int(sys.argv[2]), int(sys.argv[3])
cfg = dict(bench.CFG5); cfg["num_freq"] = nf
tel = cylinder.PolarisedCylinderTelescope.from_config(cfg)
ctx = device.get_context(workspace_bytes=40 << 30)
out = {}
for rep in range(2):
    ctx.prof_reset(2)
    torch.cuda.synchronize(); t0 = time.perf_counter()
    beam = btgen.beam_m_all(tel, ctx=ctx, max_bytes=24 << 30, m_range=(m0, m0 + cnt - 1))
    ctx.sync(); torch.cuda.synchronize(); dt = time.perf_counter() - t0
    pr = ctx.prof_report()
    out = dict(fft=os.environ.get("DM_BT_FFT", "1"), nfreq=nf, m0=m0, count=cnt, seconds=dt, classes_ms={k: round(v["ms"], 1) for k, v in pr.items() if v["ms"] > 0.05},
               checksum=float(np.abs(beam.cpu().numpy()).sum()))
    del beam
print(json.dumps(out))
