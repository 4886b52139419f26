#!/usr/bin/env python3
"""Where do the KL eigenvalues of a configs[1] block leave the oracle's?  For the given m: GPU SVD products -> GPU S, N ->
(a) GPU eigh_gen, (b) scipy eigh on the GPU's S, N, (c) the oracle's chain on the same block."""
import os, sys, tempfile
import numpy as np
import scipy.linalg as la
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench
from driftscan_amd import btgen, device
from oracle import kl as okl, svdchain as osvd

ms = [int(x) for x in sys.argv[1:]] or [102, 104, 96, 64]
ctx = device.get_context(workspace_bytes=24 << 30)
tel, bt, kl = bench.build_objects(tempfile.mkdtemp())
beam_all = btgen.beam_m_all(tel, ctx=ctx)
allm = list(range(tel.mmax + 1))
for compact in (True, False):
    res = bt.svd_device(beam_all, ms=allm if compact else None)
    sv_all = res["singularvalues"].cpu().numpy()
    for mi in allm:
        bt._dev[mi] = dict(beam_svd=res["beam_svd"][mi], beam_ut=res["beam_ut"][mi], singularvalues=sv_all[mi])
        bt._sv_host[mi] = sv_all[mi]
    bt.__dict__.pop("_stack_memo", None)
    S, N, ndofs, off = kl.sn_covariance_device(ms)
    ctx.sync()
    Sh, Nh = S.cpu().numpy(), N.cpu().numpy()
    out_alone = [kl._transform_batch([mi], to_host=True)[0] for mi in ms]
    out_batch = kl._transform_batch(ms, to_host=True)
    out_all = kl._transform_batch(allm, to_host=True)
    noisew = bt._noisew()[:, : tel.nbase]
    for i, mi in enumerate(ms):
        n = int(ndofs[i])
        Sm = Sh[off[i]: off[i] + n * n].reshape(n, n); Nm = Nh[off[i]: off[i] + n * n].reshape(n, n)
        ev_sc = la.eigh(Sm, Nm, eigvals_only=True)
        blk = beam_all[mi].cpu().numpy()
        ref = osvd.svd_m(blk, noisew, polsvcut=bt.polsvcut)
        cs, cn = okl.sn_covariance(ref["beam_svd"], ref["beam_ut"], ref["singularvalues"], kl.signal(), kl.foreground(), kl._npower(1.0), svcut=bt.svcut)
        ev_o = okl.kl_transform_m(cs, cn)[0]
        lam = np.abs(ev_o).max()
        # invariants of S, N between the two bases
        evN_g, evN_o = np.linalg.eigvalsh(Nm), np.linalg.eigvalsh(cn)
        evS_g, evS_o = np.linalg.eigvalsh(Sm), np.linalg.eigvalsh(cs)
        print("compact %d m %3d ndof %4d | gpu-eigh vs scipy(S_gpu,N_gpu): alone %.1e batch %.1e all %.1e | scipy(S_gpu,N_gpu) vs oracle %.1e | "
              "spec(N) gpu vs oracle %.1e  spec(S) %.1e | cond(N) %.1e lam_max %.2e herm(S) %.1e herm(N) %.1e"
              % (compact, mi, n, np.abs(out_alone[i][0] - ev_sc).max() / lam, np.abs(out_batch[i][0] - ev_sc).max() / lam,
                 np.abs(out_all[mi][0] - ev_sc).max() / lam, np.abs(ev_sc - ev_o).max() / lam,
                 np.abs(evN_g - evN_o).max() / np.abs(evN_o).max(), np.abs(evS_g - evS_o).max() / np.abs(evS_o).max(),
                 evN_o.max() / evN_o.min(), lam, np.abs(Sm - Sm.conj().T).max() / np.abs(Sm).max(), np.abs(Nm - Nm.conj().T).max() / np.abs(Nm).max()), flush=True)
