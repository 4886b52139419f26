#!/usr/bin/env python3
"""One-GPU probe of BASELINE configs[2] (128-feed polarised cylinder, nfreq = 64, lmax = mmax = 512).

The full job is 513 m-blocks sharded over 8 GPUs (64-65 blocks per GPU).  This script runs a SAMPLE of
it on one GPU and writes what a block costs, stage by stage, so that the 8-GPU wall time can be
projected from measured numbers:

  * BT-gen for the m range [m0, m0 + nm) over ALL (frequency, baseline) columns (map synthesis and the
    ring DFT are per column, independent of how many m are kept; the Legendre stage is per m),
  * SVD chain + pinv on each sampled block,
  * covariance projections + generalised eigenproblem (KLTransform) on each sampled block,

and checks the size-independent properties the full-size parity tests use (row orthogonality of
beam_ut, beam_svd . invbeam_svd = I, E N E^H = I, E S E^H = diag(lambda)).

    python scratch/config3_probe.py --m 100 200 --nm-bt 4 --out gpurun_out/config3_probe.json
"""
import argparse
import json
import os
import sys
import tempfile
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

CFG3 = dict(num_freq=64, freq_start=400.0, freq_end=500.0, freq_mode="edge", num_cylinders=4, cylinder_width=12.0,
            num_feeds=16, feed_spacing=0.4, tsys=1.0, force_lmax=512, force_mmax=512)


# BASELINE configs[4]: CHIME-like stress case: 4 cylinders x 64 dual-pol feeds, 256 channels
CFG5 = dict(num_freq=256, freq_start=400.0, freq_end=800.0, freq_mode="edge", num_cylinders=4, cylinder_width=14.5,
            num_feeds=64, feed_spacing=0.3, tsys=1.0, force_lmax=1024, force_mmax=1024)


def log(*a):
    print(time.strftime("%H:%M:%S"), *a, flush=True)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--m", type=int, nargs="+", default=[100])
    ap.add_argument("--nm-bt", type=int, default=1, help="m-blocks per BT-gen call (timing of the m-dependent part)")
    ap.add_argument("--bt-gb", type=float, default=48.0, help="scratch budget of BT-gen (maps + ring DFT)")
    ap.add_argument("--config", type=int, default=3, choices=[3, 5], help="BASELINE configs[2] (3) or configs[4] (5)")
    ap.add_argument("--nfreq", type=int, default=0, help="reduce for a quick plumbing run (0 = the config's own)")
    ap.add_argument("--workspace-gb", type=int, default=32)
    ap.add_argument("--skip-pinv", action="store_true")
    ap.add_argument("--lapack", action="store_true", help="with --checks: scipy.linalg.eigh(S, N) on the host for the same "
                    "pencil — its eigenvalues and ITS residual E N E^H - I, the yardstick for ours")
    ap.add_argument("--kl-class", action="store_true", help="KL through KLTransform._transform_batch (the product path, with its "
                    "own memory management) instead of the bare projections + eigh_gen calls")
    ap.add_argument("--kl-fresh-gb", type=int, default=0, help="before eigh_gen: drop every other device buffer and open a fresh "
                    "context with a workspace of this many GB (configs[4]: n = 32 576 needs ~140 GB in one arena)")
    ap.add_argument("--svd-once", action="store_true", help="time the first SVD call only (no separate allocator warm-up call)")
    ap.add_argument("--svd-batch", type=int, default=1, help="m-blocks [m0, m0 + n) pushed through the SVD chain in one call")
    ap.add_argument("--skip-kl", action="store_true")
    ap.add_argument("--checks", action="store_true", help="host-side property checks (downloads the products)")
    ap.add_argument("--out", default="gpurun_out/config3_probe.json")
    args = ap.parse_args()

    import torch

    from driftscan_amd import beamtransfer, btgen, cylinder, device, kltransform

    cfg = dict(CFG3 if args.config == 3 else CFG5)
    if args.nfreq:
        cfg["num_freq"] = args.nfreq
    tel = cylinder.PolarisedCylinderTelescope.from_config(cfg)
    log("telescope: nfreq %d nbase %d ntel %d lmax %d mmax %d" % (tel.nfreq, tel.nbase, 2 * tel.nbase, tel.lmax, tel.mmax))
    ctx = device.get_context(workspace_bytes=args.workspace_gb << 30)
    res = dict(config=cfg, nbase=int(tel.nbase), lmax=int(tel.lmax), mmax=int(tel.mmax), blocks=[])
    os.makedirs(os.path.dirname(os.path.abspath(args.out)), exist_ok=True)

    def sync():
        ctx.sync()
        torch.cuda.synchronize()

    with tempfile.TemporaryDirectory() as tmp:
        bt = beamtransfer.BeamTransfer(tmp, telescope=tel)
        kl = kltransform.KLTransform.from_config(dict(threshold=0.1), bt, subdir="kl")
        for m0 in args.m:
            rec = dict(m=m0)
            # ---- BT-gen ------------------------------------------------------------------
            for nm in sorted({args.svd_batch, args.nm_bt}):
                m1 = min(m0 + nm - 1, tel.mmax)
                sync()
                t0 = time.perf_counter()
                blk = btgen.beam_m_all(tel, ctx=ctx, max_bytes=int(args.bt_gb * (1 << 30)), m_range=(m0, m1))
                sync()
                rec["btgen_s_nm%d" % (m1 - m0 + 1)] = time.perf_counter() - t0
                log("m %d: BT-gen of %d block(s): %.2f s" % (m0, m1 - m0 + 1, rec["btgen_s_nm%d" % (m1 - m0 + 1)]))
                if nm != args.svd_batch:
                    del blk
                else:
                    beam = blk
                torch.cuda.empty_cache()
            rec["beam_absmax"] = float(beam.abs().max().item())
            # ---- SVD chain ---------------------------------------------------------------
            # first call: the workspace arena and torch's caching allocator grow to the size of this problem
            # (hipMalloc of tens of GB takes seconds); a pipeline pays that once per process, not per batch
            ctx.prof_reset(True)
            sync()
            t0 = time.perf_counter()
            out = bt.svd_device(beam, skip_svd_inv=args.skip_pinv)
            sync()
            rec["svd_first_call_s"] = time.perf_counter() - t0
            if not args.svd_once:
                del out
                ctx.prof_reset(True)
                sync()
                t0 = time.perf_counter()
                out = bt.svd_device(beam, skip_svd_inv=args.skip_pinv)
                sync()
            sv = out["singularvalues"].cpu().numpy()
            rec["svd_s"] = time.perf_counter() - t0
            rec["svd_kernels_ms"] = {k: round(v["ms"], 1) for k, v in ctx.prof_report().items()}
            rec["svd_kernels_tflops"] = {k: round(v["flops"] / (v["ms"] * 1e-3) / 1e12, 1) for k, v in ctx.prof_report().items()
                                         if v["ms"] > 0 and not k.startswith("trd")}
            log("m %d: SVD kernel classes (ms): %s" % (m0, rec["svd_kernels_ms"]))
            rec["svd_blocks"] = int(beam.shape[0])
            bt._dev[m0] = dict(beam_svd=out["beam_svd"][0], beam_ut=out["beam_ut"][0], singularvalues=sv[0])
            svnum, _ = bt._svd_num(m0)
            rec["ndof"] = int(svnum.sum())
            rec["svnum_min_max"] = [int(svnum.min()), int(svnum.max())]
            log("m %d: SVD chain of %d block(s) %.2f s (first call, growing the arenas: %.2f s), ndof %d (modes per frequency %d..%d of %d)"
                % (m0, rec["svd_blocks"], rec["svd_s"], rec["svd_first_call_s"], rec["ndof"], svnum.min(), svnum.max(),
                   sv.shape[-1]))
            if args.checks and not args.skip_pinv:
                ut = out["beam_ut"][0].cpu().numpy()
                bs = out["beam_svd"][0].cpu().numpy()
                ib = out["invbeam_svd"][0].cpu().numpy()
                worst_u = worst_p = 0.0
                noisew = bt._noisew()
                for fi in range(0, tel.nfreq, max(1, tel.nfreq // 4)):
                    n = int(svnum[fi])                      # modes above svcut (kappa <= 1/svcut)
                    if n == 0:
                        continue
                    u = ut[fi, :n] / noisew[fi][None, :]     # beam_ut = ut * noisew (beamtransfer.py:877)
                    worst_u = max(worst_u, float(np.abs(u @ u.conj().T - np.eye(n)).max()))
                    b2 = bs[fi, :n].reshape(n, -1)
                    i2 = ib[fi].reshape(-1, ib.shape[-1])[:, :n]
                    worst_p = max(worst_p, float(np.abs(b2 @ i2 - np.eye(n)).max()))
                rec["check_ut_orth"] = worst_u
                rec["check_beam_pinv"] = worst_p
                log("m %d: |U U^H - I| %.2e, |beam_svd invbeam_svd - I| %.2e" % (m0, worst_u, worst_p))
            del out, beam
            torch.cuda.empty_cache()
            json.dump(res | dict(blocks=res["blocks"] + [rec]), open(args.out, "w"), indent=1)
            # ---- KL ----------------------------------------------------------------------
            if not args.skip_kl and rec["ndof"] > 0 and args.kl_class:
                sync()
                t0 = time.perf_counter()
                r = kl._transform_batch([m0], to_host=False)[0]
                sync()
                rec["kl_class_s"] = time.perf_counter() - t0
                evk = r[0].cpu().numpy()
                rec["kl_nkept"] = int((evk >= kl.threshold).sum())
                rec["kl_evals_min_max"] = [float(evk.min()), float(evk.max())]
                rec["hbm_peak_gb"] = torch.cuda.max_memory_allocated() / 2 ** 30
                rec["workspace_gb"] = device.get_context().lib.dm_ctx_workspace_bytes(device.get_context().h) / 2 ** 30
                log("m %d: KLTransform._transform_batch %.2f s (with the C_l tables of the first call), n = %d, kept %d modes, "
                    "arena %.0f GB" % (m0, rec["kl_class_s"], evk.size, rec["kl_nkept"], rec["workspace_gb"]))
                res["blocks"].append(rec)
                json.dump(res, open(args.out, "w"), indent=1)
                continue
            if not args.skip_kl and rec["ndof"] > 0:
                ctx.prof_reset(True)
                sync()
                t0 = time.perf_counter()
                S, N, ndofs, off = kl.sn_covariance_device([m0])
                sync()
                rec["kl_cov_first_s"] = time.perf_counter() - t0   # includes the host-side C_l model tables (once per job)
                del S, N
                ctx.prof_reset(True)
                t0 = time.perf_counter()
                S, N, ndofs, off = kl.sn_covariance_device([m0])
                sync()
                rec["kl_cov_s"] = time.perf_counter() - t0
                pr = ctx.prof_report()
                gz = pr.get("zgemm_grouped") or {}
                rec["kl_cov_zgemm_ms"] = gz.get("ms")
                rec["kl_cov_zgemm_tflops"] = (gz.get("flops", 0.0) / (gz["ms"] * 1e-3) / 1e12) if gz.get("ms") else None
                log("m %d: covariance projections %.3f s (first call %.2f s with the C_l tables); grouped ZGEMM %.1f ms = %.1f TFLOP/s"
                    % (m0, rec["kl_cov_s"], rec["kl_cov_first_s"], rec["kl_cov_zgemm_ms"] or 0.0,
                       rec["kl_cov_zgemm_tflops"] or 0.0))
                ctx.prof_reset(True)
                if args.checks:
                    n = int(ndofs[0])
                    Sh = S[: n * n].cpu().numpy().reshape(n, n)
                    Nh = N[: n * n].cpu().numpy().reshape(n, n)
                if args.kl_fresh_gb:
                    bt._dev.pop(m0, None)
                    beamtransfer.BeamTransfer._clcache.clear()
                    kl._cvsg = kl._cvfg = None
                    torch.cuda.empty_cache()
                    device.reset_context()
                    ctx = device.get_context(workspace_bytes=args.kl_fresh_gb << 30)
                    ctx.prof_reset(True)
                    log("m %d: fresh context, workspace %d GB, torch holds %.1f GB" % (m0, args.kl_fresh_gb,
                                                                                     torch.cuda.memory_allocated() / 2 ** 30))
                t0 = time.perf_counter()
                cut = ("upper", kl.threshold)
                evals, evoff, evecs, ac, _ = ctx.eigh_gen(S, N, ndofs, off, cut=cut)
                sync()
                rec["kl_eigh_s"] = time.perf_counter() - t0
                ev = evals.cpu().numpy()[: int(ndofs[0])]
                rec["kl_nkept"] = int((ev >= kl.threshold).sum())
                rec["kl_evals_min_max"] = [float(ev.min()), float(ev.max())]
                rec["kl_add_const"] = float(ac[0])
                rec["kernels_ms"] = {k: v["ms"] for k, v in ctx.prof_report().items()}
                log("m %d: eigh_gen %.2f s, n = %d, kept %d modes, add_const %g"
                    % (m0, rec["kl_eigh_s"], int(ndofs[0]), rec["kl_nkept"], rec["kl_add_const"]))
                if args.checks and rec["kl_nkept"] > 0:
                    n, nk = int(ndofs[0]), rec["kl_nkept"]
                    pick = np.arange(n - nk, n)
                    if nk > 256:  # a sample of the kept modes is enough at this size (the check is O(nk n^2) on the host)
                        pick = np.unique(np.linspace(n - nk, n - 1, 256).astype(np.int64))
                        nk = pick.size
                    E = evecs[: n * n].view(n, n)[torch.as_tensor(pick, device=evecs.device)].cpu().numpy()  # evals ascend
                    lam = ev[pick]
                    EN = E @ Nh @ E.conj().T
                    ES = E @ Sh @ E.conj().T
                    rec["check_ENE"] = float(np.abs(EN - np.eye(nk)).max())
                    rec["check_ESE_offdiag"] = float(np.abs(ES - np.diag(np.diag(ES))).max() / np.abs(ES).max())
                    rec["check_ESE_diag"] = float(np.abs(np.sort(np.diag(ES).real) - np.sort(lam)).max() / np.abs(lam).max())
                    log("m %d: |E N E^H - I| %.2e, offdiag(E S E^H)/max %.2e, diag vs lambda %.2e"
                        % (m0, rec["check_ENE"], rec["check_ESE_offdiag"], rec["check_ESE_diag"]))
                    if args.lapack:
                        import scipy.linalg as la

                        t0 = time.perf_counter()
                        lev, lV = la.eigh(Sh, Nh)
                        rec["lapack_s"] = time.perf_counter() - t0
                        lE = lV.T.conj()[pick]
                        rec["lapack_ENE"] = float(np.abs(lE @ Nh @ lE.conj().T - np.eye(nk)).max())
                        rec["evals_vs_lapack"] = float(np.abs(ev - lev).max() / np.abs(lev).max())
                        rec["cond_N"] = float(np.linalg.cond(Nh))
                        log("m %d: LAPACK zhegvd on the host (%.0f s): ITS |E N E^H - I| %.2e; eigenvalues ours vs LAPACK %.2e "
                            "of lambda_max; cond(N) %.2e" % (m0, rec["lapack_s"], rec["lapack_ENE"], rec["evals_vs_lapack"],
                                                             rec["cond_N"]))
                del S, N, evecs
                torch.cuda.empty_cache()
            rec["hbm_peak_gb"] = torch.cuda.max_memory_allocated() / 2 ** 30
            rec["workspace_gb"] = ctx.lib.dm_ctx_workspace_bytes(ctx.h) / 2 ** 30
            res["blocks"].append(rec)
            bt._dev.pop(m0, None)
            json.dump(res, open(args.out, "w"), indent=1)
    log("done")


if __name__ == "__main__":
    main()
