import os, sys
os.environ["DM_TRD_TWOSTAGE"] = "1"
os.environ["DM_SB_DUMP"] = "/tmp/sbdump"
sys.path.insert(0, os.path.join(os.path.dirname(__file__), ".."))
import numpy as np
from driftscan_amd._lib import Context
ctx = Context(0, workspace_bytes=2 << 30)
rng = np.random.default_rng(5)
n = int(sys.argv[1]) if len(sys.argv) > 1 else 97
nb = int(sys.argv[2]) if len(sys.argv) > 2 else 1
A = rng.standard_normal((nb, n, n)) + 1j * rng.standard_normal((nb, n, n))
A = A + A.conj().transpose(0, 2, 1)
for rep in range(2):
    ev, W = ctx.herm_eig(ctx.to_device(np.triu(A)), n, n, strideC=n * n, batch=nb)
    ctx.sync()
print("ok", np.abs(np.sort(ev.cpu().numpy().reshape(nb, n), axis=1) - np.linalg.eigvalsh(A)).max())
dbg = np.fromfile("/tmp/sbdump.dbg", dtype=np.uint64).reshape(-1, 2).astype(np.int64)
t0 = dbg[dbg[:, 0] > 0, 0].min()
st = (dbg[: n - 1, 0] - t0) / 100.0
en = (dbg[: n - 1, 1] - t0) / 100.0
print("total us", en.max())
for s_ in list(range(0, 24)) + list(range(24, n - 1, max(1, n // 24))):
    ntask = (n - s_ - 2) // 32 + 1
    print("sweep %4d start %9.1f end %9.1f dur %7.1f tasks %3d us/task %.2f lag %.2f" % (s_, st[s_], en[s_], en[s_] - st[s_], ntask, (en[s_] - st[s_]) / ntask, st[s_] - st[s_ - 1] if s_ else 0))
