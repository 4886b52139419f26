import os, sys
os.environ["DM_TRD_TWOSTAGE"] = "1"
os.environ["DM_SB_DUMP"] = "/tmp/sbdump"
sys.path.insert(0, os.path.join(os.path.dirname(__file__), ".."))
import numpy as np
from driftscan_amd._lib import Context
ctx = Context(0, workspace_bytes=1 << 30)
rng = np.random.default_rng(5)
n = int(sys.argv[1]) if len(sys.argv) > 1 else 97
nb = int(sys.argv[2]) if len(sys.argv) > 2 else 1
A = rng.standard_normal((nb, n, n)) + 1j * rng.standard_normal((nb, n, n))
A = A + A.conj().transpose(0, 2, 1)
try:
    ev, W = ctx.herm_eig(ctx.to_device(np.triu(A)), n, n, strideC=n * n, batch=nb)
    ctx.sync()
    print("ok", np.abs(np.sort(ev.cpu().numpy().reshape(nb, n), axis=1) - np.linalg.eigvalsh(A)).max())
except Exception as e:
    print("ERR", e)
    prog = np.fromfile("/tmp/sbdump.prog", dtype=np.uint32)
    nxt = np.fromfile("/tmp/sbdump.next", dtype=np.int32)
    print("prog", prog[:n].tolist())
    print("next/owner/qhead/err", nxt.tolist())
    dbg = np.fromfile("/tmp/sbdump.dbg", dtype=np.uint64).reshape(-1, 2)
    t0 = dbg[dbg[:, 0] > 0, 0].min()
    for s_ in range(min(16, n)):
        print("sweep", s_, "start %.1f us end %.1f us" % ((int(dbg[s_, 0]) - int(t0)) / 100.0, (int(dbg[s_, 1]) - int(t0)) / 100.0))
