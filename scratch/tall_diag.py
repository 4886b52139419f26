#!/usr/bin/env python3
"""Singular values of a configs[2] block (3 frequencies) from the device chain against the oracle; run with DM_SVD_TALL=0/1."""
import os, sys, tempfile
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench
from driftscan_amd import beamtransfer, btgen, cylinder, device
from oracle import svdchain as osvd
m = int(sys.argv[1])
ctx = device.get_context(workspace_bytes=40 << 30)
tel = cylinder.PolarisedCylinderTelescope.from_config(dict(bench.CFG3))
bt = beamtransfer.BeamTransfer(tempfile.mkdtemp(), telescope=tel)
beam = btgen.beam_m_all(tel, ctx=ctx, max_bytes=24 << 30, m_range=(m, m))
res = bt.svd_device(beam, ms=[m])
ctx.sync()
sv = res["singularvalues"][0].cpu().numpy()
nm = res["nmodes"][0]
fs = [0, 31, 63]
blk = beam[0][fs].cpu().numpy()
noisew = bt._noisew()[:, : tel.nbase]
o = osvd.svd_m(blk, noisew[fs], polsvcut=bt.polsvcut)
for k, f in enumerate(fs):
    so = o["singularvalues"][k]; sg = sv[f]
    print("m %d f %2d: nmodes gpu %d oracle %d | sigma_max gpu %.6e oracle %.6e | max|dsigma|/sigma_max %.2e | sum gpu %.6e oracle %.6e | #>1e-6: gpu %d oracle %d"
          % (m, f, nm[f], o["nmodes"][k], sg.max(), so.max(), np.abs(sg - so).max() / so.max(), sg.sum(), so.sum(),
             (sg > 1e-6 * sg.max()).sum(), (so > 1e-6 * so.max()).sum()))
    print("    gpu   ", np.array2string(sg[:8], precision=4), "...", np.array2string(sg[nm[f]-3:nm[f]], precision=3))
    print("    oracle", np.array2string(so[:8], precision=4), "...", np.array2string(so[o['nmodes'][k]-3:o['nmodes'][k]], precision=3))
