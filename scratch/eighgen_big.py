"""`eigh_gen` at CHIME size (configs[4]: ndof = 32 576 at m = 300, about half of the modes kept): seconds per call with the
one-stage and the two-stage tridiagonalisation, residuals on a sample of the kept modes.  Matrices are made on the device."""
import os, sys, time
sys.path.insert(0, os.path.join(os.path.dirname(__file__), ".."))
import numpy as np
import torch
from driftscan_amd import device
from driftscan_amd._lib import block_offsets

n = int(sys.argv[1]) if len(sys.argv) > 1 else 32576
modes = sys.argv[2].split(",") if len(sys.argv) > 2 else ["1", "0"]
ctx = device.get_context(workspace_bytes=int(float(os.environ.get("WS_GB", "150")) * (1 << 30)))
g = torch.Generator(device="cuda").manual_seed(8)


def rnd(rows, cols):
    return torch.view_as_complex(torch.randn((rows, cols, 2), generator=g, device="cuda", dtype=torch.float64))


def make():
    r = int(n * 17234 / 32576)   # the signal covariance has this many modes above the cut (configs[4], m = 300: 17 234)
    X = rnd(n, r)
    d = torch.logspace(2, 0, r, device="cuda", dtype=torch.float64)
    S = (X * d) @ X.conj().T
    del X
    Y = rnd(n, n // 8)
    N = Y @ Y.conj().T / (n // 8) + torch.eye(n, device="cuda", dtype=torch.complex128)
    del Y
    return S.contiguous(), N.contiguous()


for mode in modes:
    os.environ["DM_TRD_TWOSTAGE"] = mode
    S, N = make()
    # keep the modes with eigenvalue >= the median (KLTransform keeps S/N >= threshold: 17 234 of 32 576 at configs[4])
    thr = 0.0
    off, tot = block_offsets([n])
    torch.cuda.synchronize()
    ctx.prof_reset(True)
    t0 = time.time()
    ev, evoff, E, ac, _ = ctx.eigh_gen(S.reshape(-1), N.reshape(-1), [n], off, cut=("upper", 1e-2))
    ctx.sync(); torch.cuda.synchronize()
    dt = time.time() - t0
    nk = int(ctx.last_nkeep[0])
    del S, N
    torch.cuda.empty_cache()
    S, N = make() if False else (None, None)
    print({k: (round(v["ms"] / 1e3, 2), round(v["flops"] / max(v["ms"], 1e-9) / 1e9, 1)) for k, v in ctx.prof_report().items()}, flush=True)
    print("n %d twostage %s: %.2f s, kept %d modes, ac %g, evals %.3e .. %.3e" % (n, mode, dt, nk, ac[0], float(ev[0]), float(ev[-1])), flush=True)
    del ev, E
    torch.cuda.empty_cache()
