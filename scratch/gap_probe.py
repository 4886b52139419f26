import sys, os, tempfile, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import bench, torch
from driftscan_amd import device, btgen
from driftscan_amd._lib import block_offsets
ctx = device.get_context(workspace_bytes=24 << 30)
def T(label, t0):
    torch.cuda.synchronize(); ctx.sync()
    t = time.perf_counter(); print("%-28s %8.2f ms" % (label, 1e3 * (t - t0))); return time.perf_counter()
with tempfile.TemporaryDirectory() as tmp:
    tel, bt, kl = bench.build_objects(tmp)
    bench.hot_path_step(tel, bt, kl, ctx)
    for rep in range(2):
        torch.cuda.synchronize(); t = time.perf_counter()
        beam_all = btgen.beam_m_all(tel, ctx=ctx); t = T("btgen", t)
        res = bt.svd_device(beam_all); t = T("svd", t)
        ms = list(range(tel.mmax + 1))
        sv = res["singularvalues"].cpu().numpy(); t = T("sv cpu", t)
        bt._dev = {mi: dict(beam_svd=res["beam_svd"][mi], beam_ut=res["beam_ut"][mi], singularvalues=sv[mi]) for mi in ms}
        t = T("dict", t)
        batch = list(kl._batches(ms))[0]; t = T("batches", t)
        ndofs = np.array([int(bt.ndof(mi)) for mi in ms], dtype=np.int64); t = T("ndofs", t)
        off, tot = block_offsets(ndofs)
        S = ctx.empty((max(tot, 1),), np.complex128); N = ctx.empty((max(tot, 1),), np.complex128); t = T("empty", t)
        sig = kl.signal(); fg = kl.foreground(); t = T("signal/fg", t)
        bt.project_matrix_sky_to_svd_device(ms, sig, S, off); t = T("proj S", t)
        bt.project_matrix_sky_to_svd_device(ms, fg, N, off); t = T("proj N", t)
        ctx.regularise(N, ndofs, off, kl._foreground_regulariser); t = T("regularise", t)
        but = torch.stack([bt._dev_products(mi)["beam_ut"] for mi in ms]); t = T("stack", t)
        svnum = np.stack([bt._svd_num(mi)[0] for mi in ms]); t = T("svnum", t)
        npw = ctx.to_device(kl._npower(1.0)); t = T("npower", t)
        ctx.project_diag(but, svnum, npw, N, off, alpha=1.0, accumulate=True); t = T("proj diag", t)
        out = ctx.eigh_gen(S, N, ndofs, off); t = T("eigh_gen", t)
        print("ndof stats", ndofs.min(), ndofs.mean(), ndofs.max(), (ndofs.astype(float)**3).sum())
