import numpy as np, sys
sys.path.insert(0, "/root/repo")
from driftscan_amd._lib import Context
ctx = Context(0)
for n in (80, 200, 500, 1000):
    rng = np.random.default_rng(n)
    X = rng.standard_normal((n, n)) + 1j * rng.standard_normal((n, n))
    lam = 10.0 ** rng.uniform(-12, 0, n)
    Q = np.linalg.qr(X)[0]
    C = (Q * lam) @ Q.conj().T
    C = 0.5 * (C + C.conj().T)
    import time
    dC = ctx.to_device(C[None])
    t0 = time.time()
    ev, W, sw = ctx.jacobi_herm(dC, n, n)
    dt = time.time() - t0
    ev = np.sort(ev.cpu().numpy()[0, :n]); ref = np.linalg.eigvalsh(C)
    W = W.cpu().numpy()[0]
    print("n", n, "sweeps", sw, "time", round(dt, 3), "err/scale", np.abs(ev - ref).max() / ref.max(), "unit", np.abs(W @ W.conj().T - np.eye(n)).max())
