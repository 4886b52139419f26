#!/usr/bin/env python3
"""One GPU's share of BASELINE configs[2] (128-feed polarised cylinder, nfreq = 64, lmax = mmax = 512,
513 m-blocks sharded over 8 GPUs), measured the way the multi-rank pipeline splits the work
(DESIGN.md section 6):

  * BT-gen: a rank transforms a CONTIGUOUS range of 65 m (map synthesis and ring DFT are per (f, b)
    column whatever the number of m kept, so few large calls are the cheap way): timed here as
    `--bt-calls` calls of 65 / bt-calls blocks each, at the low-m end (the expensive end);
  * SVD chain + pinv and KL (covariance projections + generalised eigenproblem): cost-balanced
    assignment of m to ranks; timed here on 72 blocks spread evenly over the m range, in groups of
    `--group` contiguous blocks (the inputs of those groups are generated untimed).

m-blocks are independent and there is no data-path collective, so BT-gen + SVD + KL of this share IS the
compute wall time of the 8-GPU job (products stay in HBM; file output is not timed).

    python scratch/config3_share.py --out gpurun_out/config3_share.json
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

from scratch.config3_probe import CFG3, log  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--ranks", type=int, default=8, help="GPUs of the full job")
    ap.add_argument("--group", type=int, default=8, help="contiguous m-blocks per BT-gen / SVD / KL call")
    ap.add_argument("--bt-calls", type=int, default=1, help="BT-gen calls for the rank's contiguous range of m")
    ap.add_argument("--bt-gb", type=float, default=48.0)
    ap.add_argument("--workspace-gb", type=int, default=80)
    ap.add_argument("--config4", action="store_true", help="BASELINE configs[3]: add the DoubleKL filter and the exact Fisher "
                    "matrix (PSExact, analytic stand-in bands) of every sampled block")
    ap.add_argument("--limit-groups", type=int, default=0, help="stop after this many groups (0 = the whole share)")
    ap.add_argument("--out", default="gpurun_out/config3_share.json")
    args = ap.parse_args()

    import torch

    from driftscan_amd import beamtransfer, btgen, cylinder, device, kltransform

    tel = cylinder.PolarisedCylinderTelescope.from_config(dict(CFG3))
    nm_total = tel.mmax + 1
    share = (nm_total + args.ranks - 1) // args.ranks            # 65 blocks for the largest share
    ngroups = (share + args.group - 1) // args.group
    # group g covers m in [g * stride, g * stride + group): the share samples the whole m range evenly
    stride = nm_total // ngroups
    groups = [(g * stride, min(g * stride + args.group, nm_total) - 1) for g in range(ngroups)]
    if args.limit_groups:
        groups = groups[: args.limit_groups]
    log("telescope: nfreq %d nbase %d lmax %d; %d m-blocks in total, share of one of %d GPUs: %d blocks in %d groups"
        % (tel.nfreq, tel.nbase, tel.lmax, nm_total, args.ranks, sum(b - a + 1 for a, b in groups), len(groups)))
    # the arena is sized once (growing it later costs a hipMalloc of tens of GB, seconds each time)
    ctx = device.get_context(workspace_bytes=args.workspace_gb << 30)

    def sync():
        ctx.sync()
        torch.cuda.synchronize()

    res = dict(config=dict(CFG3), ranks=args.ranks, groups=[], totals={})
    os.makedirs(os.path.dirname(os.path.abspath(args.out)), exist_ok=True)
    with tempfile.TemporaryDirectory() as tmp:
        bt = beamtransfer.BeamTransfer(tmp, telescope=tel)
        kl = kltransform.KLTransform.from_config(dict(threshold=0.1), bt, subdir="kl")
        dk = ps = None
        tdk = tps = 0.0
        if args.config4:
            from driftscan_amd import doublekl, psestimation

            dk = doublekl.DoubleKL.from_config(dict(threshold=0.1, foreground_threshold=100.0), bt, subdir="dk")
            dk._cvsg, dk._cvfg = None, None
            ps = psestimation.PSExact.from_config(
                dict(bandtype="polar", num_theta=3, threshold=0.1,
                     k_bands=[dict(spacing="linear", start=0.0, stop=0.25, num=4)]), kl, subdir="ps")
            ps.genbands()
            log("config 4: DoubleKL (foreground_threshold 100) + PSExact with %d bands" % ps.nbands)
        t0 = time.perf_counter()
        kl.signal(); kl.foreground()
        bt._cl_device(kl.signal()); bt._cl_device(kl.foreground())
        sync()
        res["cl_tables_s"] = time.perf_counter() - t0   # host-side C_l(nu, nu') model tables, once per job
        log("C_l tables: %.1f s" % res["cl_tables_s"])
        ctx.prof_reset(True)
        tall = time.perf_counter()
        tb = ts = tk = 0.0
        nblk = 0
        # ---- BT-gen of a rank's contiguous range (rank 0: the lowest m, the longest Legendre sums)
        per_call = (share + args.bt_calls - 1) // args.bt_calls
        res["btgen_calls"] = []
        for c in range(args.bt_calls if not args.limit_groups else 1):
            m_lo, m_hi = c * per_call, min((c + 1) * per_call, share) - 1
            sync()
            t0 = time.perf_counter()
            beam = btgen.beam_m_all(tel, ctx=ctx, max_bytes=int(args.bt_gb * (1 << 30)), m_range=(m_lo, m_hi))
            sync()
            dt = time.perf_counter() - t0
            tb += dt
            res["btgen_calls"].append(dict(m_lo=m_lo, m_hi=m_hi, s=dt))
            log("BT-gen m %d..%d (%d blocks, %.1f GB): %.2f s" % (m_lo, m_hi, m_hi - m_lo + 1,
                                                                 beam.numel() * 16 / 2 ** 30, dt))
            del beam
            torch.cuda.empty_cache()   # the 110 GB of blocks go back to the driver; everything after re-uses its cache
        t_untimed = 0.0
        warm = True   # the first group runs twice: the first pass sizes torch's caching allocator (untimed)
        for (m_lo, m_hi) in [groups[0]] + groups:
            ms = list(range(m_lo, m_hi + 1))
            sync()
            t0 = time.perf_counter()
            beam = btgen.beam_m_all(tel, ctx=ctx, max_bytes=int(args.bt_gb * (1 << 30)), m_range=(m_lo, m_hi))
            sync()
            t1 = time.perf_counter()
            t_untimed += t1 - t0
            out = bt.svd_device(beam)
            sv = out["singularvalues"].cpu().numpy()
            sync()
            t2 = time.perf_counter()
            del beam
            for i, mi in enumerate(ms):
                bt._dev[mi] = dict(beam_svd=out["beam_svd"][i], beam_ut=out["beam_ut"][i], singularvalues=sv[i])
            ndofs = [int(bt.ndof(mi)) for mi in ms]
            prods = None
            for batch in kl._batches(ms):
                prods = kl._transform_batch(batch, to_host=False)
            sync()
            t3 = time.perf_counter()
            nk = int(ctx.last_nkeep.sum()) if hasattr(ctx, "last_nkeep") else None
            out_keep = out if args.config4 else None
            del out, prods
            for mi in ms:
                bt._dev.pop(mi, None)
            t_dk = t_ps = 0.0
            if args.config4:
                # DoubleKL of the same blocks (two generalised eigenproblems + re-projections; modes go to the host)
                for mi in ms:
                    bt._dev[mi] = dict(beam_svd=out_keep["beam_svd"][mi - m_lo], beam_ut=out_keep["beam_ut"][mi - m_lo],
                                       singularvalues=sv[mi - m_lo])
                sync()
                t4 = time.perf_counter()
                for batch in dk._batches(ms):
                    dk._transform_batch(batch)
                sync()
                t_dk = time.perf_counter() - t4
                # exact Fisher matrix of the KL-filtered modes: the KL products are written as the pipeline does
                # (untimed), PSExact reads the modes back and projects every band
                for batch in kl._batches(ms):
                    for mi, r in zip(batch, kl._transform_batch(batch)):
                        kl._save(mi, *r)
                sync()
                t5 = time.perf_counter()
                for batch in ps._batches(ms):
                    ps.fisher_bias_batch(batch)
                sync()
                t_ps = time.perf_counter() - t5
                for mi in ms:
                    bt._dev.pop(mi, None)
                    try:
                        os.remove(kl._evfile % mi)
                    except OSError:
                        pass
                del out_keep
            if warm:
                warm = False
                log("m %3d..%3d warm-up pass (allocators): SVD %.2f s, KL %.2f s" % (m_lo, m_hi, t2 - t1, t3 - t2))
                continue
            tdk += t_dk; tps += t_ps
            rec = dict(m_lo=m_lo, m_hi=m_hi, btgen_s=t1 - t0, svd_s=t2 - t1, kl_s=t3 - t2, ndof=ndofs, kept_last_batch=nk,
                       doublekl_s=t_dk, fisher_s=t_ps)
            res["groups"].append(rec)
            ts += t2 - t1; tk += t3 - t2
            nblk += len(ms)
            log("m %3d..%3d: (input generation %.2f s, not part of the share) SVD %.2f s, KL %.2f s%s, ndof %d..%d"
                % (m_lo, m_hi, t1 - t0, t2 - t1, t3 - t2,
                   (", DoubleKL %.2f s, Fisher %.2f s" % (t_dk, t_ps)) if args.config4 else "", min(ndofs), max(ndofs)))
            res["totals"] = dict(blocks=nblk, btgen_s=tb, svd_s=ts, kl_s=tk)
            json.dump(res, open(args.out, "w"), indent=1)
        wall = tb + (ts + tk + tdk + tps) * share / max(nblk, 1)   # BT-gen of 65 contiguous blocks + the rest scaled from the sample to 65
        pr = ctx.prof_report()
        res["kernels_ms"] = {k: v["ms"] for k, v in pr.items()}
        res["kernels_tflops"] = {k: (v["flops"] / (v["ms"] * 1e-3) / 1e12 if v["ms"] > 0 else None) for k, v in pr.items()
                                 if k not in ("trd_symv", "trd_wx")}
        res["totals"] = dict(svd_kl_blocks=nblk, btgen_blocks=share, btgen_s=tb, svd_s=ts, kl_s=tk, doublekl_s=tdk, fisher_s=tps,
                             projected_job_s=wall, m_blocks_per_s_per_gpu=share / wall,
                             m_blocks_per_s_8gpu=nm_total / wall,
                             projected_job_note="%d blocks per GPU on %d GPUs, compute only (products left in HBM): BT-gen "
                                                "of the rank's contiguous range + (SVD + KL of the %d-block sample) x %d / %d"
                                                % (share, args.ranks, nblk, share, nblk))
        res["hbm_peak_gb"] = torch.cuda.max_memory_allocated() / 2 ** 30
        res["workspace_gb"] = ctx.lib.dm_ctx_workspace_bytes(ctx.h) / 2 ** 30
        json.dump(res, open(args.out, "w"), indent=1)
        log("share done: BT-gen of %d blocks %.1f s; SVD %.1f s + KL %.1f s%s on %d blocks -> projected 8-GPU job %.0f s "
            "(%.2f m-blocks/s per GPU)" % (share, tb, ts, tk,
                                           (" + DoubleKL %.1f s + Fisher %.1f s" % (tdk, tps)) if args.config4 else "",
                                           nblk, wall, share / wall))


if __name__ == "__main__":
    main()
