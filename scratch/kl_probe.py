import sys, tempfile, numpy as np
sys.path.insert(0, "/root/repo")
import bench
from driftscan_amd import device, btgen
import scipy.linalg as la
ctx = device.get_context(workspace_bytes=8 << 30)
tmp = tempfile.mkdtemp()
tel, bt, kl = bench.build_objects(tmp)
beam_all = btgen.beam_m_all(tel, ctx=ctx)
print("beam nan", bool(beam_all.isnan().any()))
ms = [0, 1, 2, 64, 128]
res = bt.svd_device(beam_all[ms])
sv = res["singularvalues"].cpu().numpy()
print("sv nan", np.isnan(sv).any(), "sweeps", res["sweeps"], "nmodes", res["nmodes"][:, :4])
bt._dev = {mi: dict(beam_svd=res["beam_svd"][i], beam_ut=res["beam_ut"][i], singularvalues=sv[i]) for i, mi in enumerate(ms)}
for mi in ms:
    svnum, _ = bt._svd_num(mi)
    S, N = kl.sn_covariance(mi)
    print("m", mi, "ndof", S.shape[0], "svnum", svnum[:6], "S nan", np.isnan(S).any(), "N nan", np.isnan(N).any(),
          "S herm", np.abs(S - S.conj().T).max(), "Nmax", np.abs(N).max())
    try:
        w = la.eigvalsh(N); print("   N eig range", w[0], w[-1])
        ev = la.eigh(S, N, eigvals_only=True); print("   lapack top", ev[-3:])
    except Exception as e:
        print("   lapack fail", e)
    from driftscan_amd import kltransform
    try:
        ev2, E, ac = kltransform.eigh_gen(S, N)
        print("   gpu top", ev2[-3:], "ac", ac)
    except Exception as e:
        print("   gpu fail", e)
