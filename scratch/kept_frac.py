import sys, os, tempfile
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch, bench
from driftscan_amd import device, btgen
ctx = device.get_context(workspace_bytes=24 << 30)
with tempfile.TemporaryDirectory() as tmp:
    tel, bt, kl = bench.build_objects(tmp)
    beam_all = btgen.beam_m_all(tel, ctx=ctx); ctx.sync()
    res = bt.svd_device(beam_all)
    sv = res["singularvalues"].cpu().numpy()
    ms = list(range(tel.mmax + 1))
    for mi in ms: bt._dev[mi] = dict(beam_svd=res["beam_svd"][mi], beam_ut=res["beam_ut"][mi], singularvalues=sv[mi])
    out = kl._transform_batch(ms, to_host=False)
    tot = kept = 0; w3 = k3 = 0.0
    for mi, o in zip(ms, out):
        ev = o[0].cpu().numpy()
        n = ev.size; k = int((ev >= kl.threshold).sum())
        tot += n; kept += k; w3 += float(n) ** 3; k3 += float(n) ** 2 * k
        if mi % 16 == 0: print(mi, n, k, ev[-3:] if n else None)
    print("kept fraction (count) %.3f, weighted by n^2 (back-transform / TRSM cost) %.3f" % (kept / tot, k3 / w3))
