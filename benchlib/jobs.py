"""The north-star workloads of bench.py through the product's own `ProductManager.generate()`: one emulated share of the
BASELINE configs[2] / configs[3] job on this GPU (`measure_share`), the whole job on N real ranks (`measure_job`), one block of
configs[4] (`measure_configs4_block`)."""
import json
import os
import sys
import time

from .common import CFG3, CFG5, FP64_MFMA_PEAK_TFLOPS, class_table, stage_work

def job_conf(workload, toy=False, truncate=False):
    """Configuration dictionary (the reference's YAML sections) of the north-star job: BASELINE configs[2]
    (`KLTransform`) or configs[3] (+ `DoubleKL` + the exact Fisher matrix); `toy`: the same job on a toy telescope
    (rehearsals of the control flow on CPU-sized boxes and in the tests)."""
    tcfg = dict(CFG3, type="PolarisedCylinder")
    if toy:
        tcfg = dict(type="PolarisedCylinder", num_freq=4, freq_start=400.0, freq_end=440.0, freq_mode="edge", num_cylinders=2,
                    cylinder_width=2.0, num_feeds=4, feed_spacing=0.4, tsys=1.0)
    kls = [dict(type="KLTransform", name="kl", threshold=0.1)]
    conf = dict(config=dict(beamtransfers=True, kltransform=True, psfisher=False, truncate=bool(truncate),
                            device_chunk_gb=float(os.environ.get("DRIFT_BENCH_BT_GB", "48")), keep_products_gb=0.0),
                telescope=tcfg, kltransform=kls)
    if workload == "configs3":
        kls.append(dict(type="DoubleKL", name="dk", threshold=0.1, foreground_threshold=100.0))
        conf["config"]["psfisher"] = True
        conf["psfisher"] = [dict(type="Full", name="ps", klname="kl", threshold=0.1, bandtype="polar", num_theta=3,
                                 k_bands=[dict(spacing="linear", start=0.0, stop=0.25, num=4)])]
    return conf


def job_budgets(toy=False):
    """Batch budgets (GB) of a rank of the north-star job on a 288 GB card: resident beam blocks of a BT-gen range / SVD
    batch / KL batch / eigensolver arena; a toy rehearsal (several ranks on one card) takes 1 GB each."""
    if toy:
        return dict(beam=1.0, svd=1.0, kl=1.0, arena=1.0)
    return dict(beam=float(os.environ.get("DRIFT_BENCH_BEAM_GB", "72")), svd=float(os.environ.get("DRIFT_BENCH_SVD_GB", "96")),
                kl=float(os.environ.get("DRIFT_BENCH_KL_GB", "110")), arena=float(os.environ.get("DRIFTMI_WORKSPACE_GB", "100")))


def storage_io_stats():
    """Seconds the writer pipeline of this process spent where (summed over its threads), `storage.io_stats`."""
    from driftscan_amd import storage

    return {k: (round(v, 3) if isinstance(v, float) else v) for k, v in storage.io_stats().items()}


def measure_share(workload, share, files=False, share_mmax=None, truncate=False, outdir=None):
    """BASELINE configs[2] / configs[3] — the north-star job — through ProductManager.generate(): rank r of N is
    emulated in this process (`parallel.set_virtual`: its contiguous, cost-balanced range of m; no process group), so
    the share's wall time is that rank's part of the N-GPU job (m-blocks are independent, the only collective is the
    all-reduce of the Fisher matrix at the very end); rank 0 holds the lowest m — the largest matrices and the m the polar
    rings of the SHT refinement couple — and is the slowest share.  Without `files` the products stay in HBM
    (DRIFTMI_STORAGE=discard); with it they go through the writer pool to a temporary directory."""
    import tempfile

    import torch
    import yaml

    r, n = (int(x) for x in share.split("/"))
    if not files:
        os.environ["DRIFTMI_STORAGE"] = "discard"
    # Batch budgets (GB): resident beam blocks of a BT-gen range / SVD batch / KL batch / eigensolver arena.  Rounds 1-3 ran every
    # share at 125 / 48 / 48 / 80.  Now 72 / 96 / 110 / 100: 11 low-m blocks per SVD batch and per eigh_gen
    # call — a third of the lock-step launch chains, and the KL eigenproblems reach the batch sizes where the two-stage
    # tridiagonalisation pays (share 0/8: 32.2 -> 28.5 s; torch peak 123 GB + the 100 GB arena of 288; the kernels of the high-m
    # shares gain 1.7 s as well).  The beam blocks of a BT-gen range + one SVD batch + the arena must fit the card: with 125 GB of
    # beam blocks share 7/8 ran out of memory.  configs[3] (DoubleKL + Fisher) runs at the same budgets: share 0/8 42.0 -> 35.4 s,
    # torch peak 106 GB.
    from driftscan_amd import device, manager, parallel

    parallel.set_virtual(r, n)
    try:
        conf = job_conf(workload, share_mmax, truncate)
        with tempfile.TemporaryDirectory(dir=outdir) as tmp:
            conf["config"]["output_directory"] = os.path.join(tmp, "prod")
            cfile = os.path.join(tmp, "params.yaml")
            with open(cfile, "w") as fh:
                yaml.dump(conf, fh)
            pm = manager.ProductManager.from_config(cfile)
            tel, bt = pm.telescope, pm.beamtransfer
            mine = bt._my_ms()
            budgets = job_budgets(bool(share_mmax))
            bt.beam_chunk_gb, bt.svd_chunk_gb = budgets["beam"], budgets["svd"]
            for kl in pm.kltransforms.values():
                kl.kl_chunk_gb = budgets["kl"]
            ctx = device.get_context(workspace_bytes=int(budgets["arena"] * (1 << 30)))
            # the host-side C_l(nu, nu') tables are made once per job (cora's models in the reference): untimed
            t0 = time.perf_counter()
            for kl in pm.kltransforms.values():
                kl.signal(); kl.foreground()
            t_cl = time.perf_counter() - t0
            noprof = os.environ.get("DRIFT_BENCH_NOPROF") == "1"   # (what the event pairs and the idle points of the stage log cost)
            ctx.prof_reset(0 if noprof else 2)      # every kernel class of the path
            bt.stage_log = None if noprof else []   # per BT-gen range / SVD batch / KL batch: wall seconds + kernel classes (device idle at the boundaries)
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            pm.generate()
            ctx.sync()
            torch.cuda.synchronize()
            dt = time.perf_counter() - t0
            pr = ctx.prof_report()
            stages = share_stages(tel, bt, pm, mine, dt)
            nbytes = 0
            if files:
                for root, _, fl in os.walk(conf["config"]["output_directory"]):
                    nbytes += sum(os.path.getsize(os.path.join(root, f)) for f in fl)
            nm = tel.mmax + 1
            classes = class_table(pr)
            kern_s = sum(v["ms"] for v in pr.values()) * 1e-3
            cov = pr.get("zgemm_cov")
            name = "configs[2]" if workload == "configs2" else "configs[3]"
            line = {
                "metric": "m-blocks/sec (BT-gen + SVD + KL)",
                "value": len(mine) / dt,
                "unit": "m-blocks/s",
                "n_gpus": 1, "steps": 1, "warmup": 0,
                "ms_per_step": 1e3 * dt,
                "higher_is_better": True, "scaling": "strong", "vs_baseline": None, "dtype": "f64", "data": "synthetic",
                "config": {"workload": "%s: share %d/%d of the %s job (nfreq=%d, nbase=%d, lmax=mmax=%d) "
                                       "through ProductManager.generate(): m = %d..%d (%d of %d m-blocks), %s%s"
                                       % (name, r, n,
                                          "toy-telescope REHEARSAL" if share_mmax else "128-feed polarised cylinder", tel.nfreq,
                                          tel.nbase, tel.lmax,
                                          mine[0], mine[-1], len(mine), nm, "KLTransform" if workload == "configs2"
                                          else "KLTransform + DoubleKL + PSExact (9 polar bands)",
                                          ", product files written" if files else ", products left in HBM (no files)"),
                           "nfreq": tel.nfreq, "nbase": tel.nbase, "lmax": tel.lmax, "mmax": tel.mmax, "share": share,
                           "sht_iter": int(tel.sht_iter), "files": bool(files), "truncate": bool(truncate), "budgets_gb": budgets,
                           "codec": os.environ.get("DRIFTMI_H5_CODEC", "lzf") if files else None},
                "share_s": dt,
                "share_note": "wall time of rank %d of %d for m = %d..%d; the job's wall time is the MAX over the N shares "
                              "(m-blocks are independent, no data-path collective) — ONE share says nothing about which is the "
                              "slowest: all N are in profiles/*_configs2_shares.json.  C_l tables %.1f s (host, once per job) not "
                              "included" % (r, n, mine[0], mine[-1], t_cl),
                "m_range": [int(mine[0]), int(mine[-1])],
                "file_bytes": nbytes,
                "kernels_ms": {k: v["ms"] for k, v in pr.items()},
                "arena_gb_at_end": float(ctx.lib.dm_ctx_workspace_bytes(ctx.h)) / float(1 << 30),
                "classes": classes,
                "kernel_s": kern_s,
                "kernel_coverage_of_wall": kern_s / dt,
                "stages": stages,
                "zgemm_cov": None if cov is None else dict(
                    ms=cov["ms"], flop=cov["flops"], launches=cov["launches"],
                    tflops=cov["flops"] / (cov["ms"] * 1e-3) / 1e12 if cov["ms"] > 0 else None,
                    frac=cov["flops"] / (cov["ms"] * 1e-3) / 1e12 / FP64_MFMA_PEAK_TFLOPS if cov["ms"] > 0 else None,
                    note="the covariance projections (B_f o C_l) B_f'^H of the KL stage (gathered-B grouped ZGEMM), "
                         "8 M N K flops per product over the HIP-event time of its launches"),
                "hbm_peak_gb": torch.cuda.max_memory_allocated() / 2 ** 30,
                "hbm_reserved_peak_gb": torch.cuda.max_memory_reserved() / 2 ** 30,
                "alloc_retries": int(torch.cuda.memory_stats().get("num_alloc_retries", 0)),   # the caching allocator ran out, emptied its cache and asked the driver again
                "io": storage_io_stats() if files else None,
                "stage_log": None if not bt.stage_log else [dict(stage=r_["stage"], m0=r_["ms"][0] if r_["ms"] else None, n=len(r_["ms"]),
                                                                 seconds=round(r_["seconds"], 3),
                                                                 kernel_s=round(sum(v["ms"] for v in r_["classes"].values()) * 1e-3, 3))
                                                            for r_ in bt.stage_log],
                "roofline": None, "cpu_baseline": None,
            }
            del pm
            return line
    finally:
        parallel.set_virtual(None)


def share_stages(tel, bt, pm, mine, wall_s):
    """SURVEY.md section 8(d) at the north-star workload: wall seconds of the three stages of the share (from the
    product's own stage log: the device is idle at every stage boundary) and the algorithmic work W_A (Legendre), W_B
    (SVD chain), W_C (projections + eig) over them as fractions of the fp64 MFMA peak — W from the REAL svnum / ndof of
    the share's blocks and the modes actually kept — plus the kernel classes of each stage, so that the grouped ZGEMM
    seconds of the SVD chain and of the eigensolver are told apart."""
    import numpy as np

    log = bt.stage_log or []
    nkeep = {}
    kls = list(pm.kltransforms.values())
    if kls:
        full = kls[0].__dict__.get("_evals_full_mem", {})
        for mi, evf in full.items():
            nkeep[mi] = int((np.asarray(evf) >= kls[0].threshold).sum()) if kls[0].subset else int(len(evf))
    WA, WB, WC = stage_work(tel, bt, mine, nkeep=nkeep if len(nkeep) == len(mine) else None)
    if len(kls) > 1 or getattr(pm, "gen_ps", False):
        WC = None   # configs[3]: DoubleKL and the Fisher estimator run in the same downstream stage; W_C covers one KLTransform only
    out = {}
    for name, W in (("btgen", WA), ("svd", WB), ("kl", WC)):
        recs = [r for r in log if r["stage"] == name]
        secs = sum(r["seconds"] for r in recs)
        cls = {}
        for r in recs:
            for k, v in r["classes"].items():
                a = cls.setdefault(k, dict(ms=0.0, flops=0.0, launches=0))
                a["ms"] += v["ms"]; a["flops"] += v["flops"]; a["launches"] += v["launches"]
        tf = (W / secs / 1e12) if (W is not None and secs > 0) else None
        out[name] = dict(seconds=secs, calls=len(recs), work_flop=W, tflops=tf,
                         frac_of_fp64_mfma_peak=None if tf is None else tf / FP64_MFMA_PEAK_TFLOPS,
                         kernel_s=sum(v["ms"] for v in cls.values()) * 1e-3,
                         classes_ms={k: round(v["ms"], 1) for k, v in sorted(cls.items(), key=lambda kv: -kv[1]["ms"])},
                         blocks_per_call=[len(r["ms"]) for r in recs])
    out["other_s"] = wall_s - sum(out[k]["seconds"] for k in ("btgen", "svd", "kl"))
    out["note"] = ("W_A = 8 Nr Lm F B P, W_B = sum_f [svd(T, P Lm) + svd(r1, (P-1) Lm) + svd(r2, Lm) + svd(n, P Lm)] + projections "
                   "(svd(a, b) = 4 (2 max min^2 + 11 min^3), r1 = r2 = min(T, P Lm), n = the frequency's kept modes), "
                   "W_C = 8 ndof^2 Lm (1 + n_F) + 8 T sum n_f^2 + eig(ndof, nkeep): SURVEY section 8(d); seconds are wall times "
                   "between device-idle points of ProductManager.generate(); other_s = spectra collection, allocation, host")
    return out


def measure_job(workload, backend="nccl", one_gpu=False, toy=False):
    """The north-star job on N REAL ranks (this process is one of them: RANK / LOCAL_RANK / WORLD_SIZE / MASTER_* from
    the launcher): every rank takes its `_my_ms()` range of BASELINE configs[2] (configs[3]: + DoubleKL + exact Fisher)
    through `ProductManager.generate()` on its own GPU — the reference's functional test is exactly this with two MPI
    ranks (tests/test_functional.py:58-88, drift/core/manager.py:278-305) — with the spectra gathered to rank 0
    (kltransform.py:21-52) and, for configs[3], the Fisher matrix all-reduced over RCCL (psestimation.py:506-507).
    Rank 0 returns the line: per-rank seconds, max / mean, seconds inside collectives, the ranks RCCL saw."""
    import tempfile

    import numpy as np
    import torch
    import torch.distributed as dist
    import yaml

    from driftscan_amd import device, manager, parallel

    world = int(os.environ["WORLD_SIZE"])
    local = 0 if one_gpu else int(os.environ.get("LOCAL_RANK", "0"))
    os.environ["DRIFTMI_DEVICE"] = str(local)
    os.environ["DRIFTMI_STORAGE"] = "discard"     # products stay in HBM, as in the share measurements
    os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
    torch.cuda.set_device(local)
    dist.init_process_group(backend=backend)
    rank = dist.get_rank()
    dev = "cuda" if backend == "nccl" else "cpu"
    one = torch.ones(1, dtype=torch.float64, device=dev)
    dist.all_reduce(one)                           # the tensor collective of the path, over the job's backend
    seen = int(round(float(one.item())))
    conf = job_conf(workload, toy)
    # one product directory for all ranks (rank 0 makes it and broadcasts the name, as a shared file system would hold it)
    tmp = tempfile.mkdtemp() if rank == 0 else None
    tmp = parallel.bcast_object(tmp)
    try:
        conf["config"]["output_directory"] = os.path.join(tmp, "prod")
        cfile = os.path.join(tmp, "params_%d.yaml" % rank)
        with open(cfile, "w") as fh:
            yaml.dump(conf, fh)
        pm = manager.ProductManager.from_config(cfile)
        tel, bt = pm.telescope, pm.beamtransfer
        mine = bt._my_ms()
        budgets = job_budgets(toy)
        bt.beam_chunk_gb, bt.svd_chunk_gb = budgets["beam"], budgets["svd"]
        for kl in pm.kltransforms.values():
            kl.kl_chunk_gb = budgets["kl"]
        ctx = device.get_context(workspace_bytes=int(budgets["arena"] * (1 << 30)))
        for kl in pm.kltransforms.values():       # C_l tables: host, once per job, untimed (cora's models in the reference)
            kl.signal(); kl.foreground()
        ctx.prof_reset(1)
        bt.stage_log = []
        parallel.collective_stats(reset=True)
        parallel.barrier()
        torch.cuda.synchronize()
        parallel.collective_stats(reset=True)
        t0 = time.perf_counter()
        pm.generate()
        ctx.sync()
        torch.cuda.synchronize()
        t_mine = time.perf_counter() - t0          # this rank's generate(): compute + its waits inside the collectives
        cs = parallel.collective_stats()
        parallel.barrier()
        t_job = time.perf_counter() - t0           # every rank has finished
        st = {k: sum(r["seconds"] for r in bt.stage_log if r["stage"] == k) for k in ("btgen", "svd", "kl")}
        vec = torch.tensor([t_mine, cs["seconds"], float(mine[0] if mine else -1), float(mine[-1] if mine else -1),
                            float(len(mine)), st["btgen"], st["svd"], st["kl"], cs["allreduce_s"], float(cs["allreduce_calls"]),
                            t_job, torch.cuda.max_memory_allocated() / 2 ** 30], dtype=torch.float64, device=dev)
        allv = [torch.zeros_like(vec) for _ in range(world)]
        dist.all_gather(allv, vec)
        line = None
        if rank == 0:
            per = [dict(rank=i, seconds=float(v[0]), compute_s=float(v[0] - v[1]), collective_s=float(v[1]), m_lo=int(v[2]),
                        m_hi=int(v[3]), m_blocks=int(v[4]), stage_s=dict(btgen=float(v[5]), svd=float(v[6]), kl=float(v[7])),
                        allreduce_s=float(v[8]), allreduce_calls=int(v[9]), hbm_peak_gb=float(v[11])) for i, v in enumerate(allv)]
            job_s = max(float(v[10]) for v in allv)
            comp = [p_["compute_s"] for p_ in per]
            nm = tel.mmax + 1
            name = "configs[2]" if workload == "configs2" else "configs[3]"
            line = {
                "metric": "m-blocks/sec (BT-gen + SVD + KL)", "value": nm / job_s, "unit": "m-blocks/s", "n_gpus": world,
                "steps": 1, "warmup": 0, "ms_per_step": 1e3 * job_s, "higher_is_better": True, "scaling": "strong",
                "vs_baseline": None, "dtype": "f64", "data": "synthetic",
                "config": {"workload": "%s: the whole %s job (nfreq=%d, nbase=%d, lmax=mmax=%d, %d m-blocks) on %d REAL ranks through "
                                       "ProductManager.generate(), one contiguous cost-balanced m-range per rank, %s, products left "
                                       "in HBM (no files)" % (name, "toy-telescope REHEARSAL" if toy else "128-feed polarised cylinder",
                                                              tel.nfreq, tel.nbase, tel.lmax, nm, world,
                                                              "KLTransform" if workload == "configs2" else
                                                              "KLTransform + DoubleKL + PSExact (9 polar bands), Fisher all-reduce"),
                           "nfreq": tel.nfreq, "nbase": tel.nbase, "lmax": tel.lmax, "mmax": tel.mmax, "ranks": world,
                           "backend": backend, "one_gpu": bool(one_gpu), "sht_iter": int(tel.sht_iter),
                           "budgets_gb": budgets},
                "job_s": job_s, "ranks_seen_by_rccl" if backend == "nccl" else "ranks_seen_by_gloo": seen,
                "per_rank": per, "rank_seconds_max": max(comp), "rank_seconds_mean": float(np.mean(comp)),
                "imbalance_max_over_mean": max(comp) / float(np.mean(comp)),
                "collective_s_max": max(p_["collective_s"] for p_ in per),
                "collectives": "pickled spectra gathered to rank 0 over %s (svdspectrum, evals), barriers%s" % (
                    "a gloo side group" if backend == "nccl" else "gloo",
                    ", Fisher + bias all-reduce over %s" % ("RCCL" if backend == "nccl" else "gloo") if workload == "configs3" else ""),
                "note": "job_s = barrier to barrier around generate() on every rank (C_l tables made before, untimed); per rank: "
                        "seconds = its generate(), collective_s = time inside barriers / gathers / all-reduce (mostly waiting for "
                        "the slowest rank), compute_s = the difference",
                "roofline": None, "cpu_baseline": None,
            }
        del pm
    finally:
        parallel.barrier()
        if rank == 0:
            import shutil

            shutil.rmtree(tmp, ignore_errors=True)
        dist.destroy_process_group()
    return line


def measure_configs4_block(m=300, checks=True, workspace_gb=100, bt_gb=24, log=None):
    """BASELINE configs[4] (CHIME-like: 512 feeds, nfreq 256, lmax = mmax = 1024, HBM-bound per-m blocks): ONE real m-block
    through the product classes — BT-gen of the 59.6 GB block, the SVD chain of all 256 frequencies (the library slices
    them), the KL transform of the block (ndof ~32 600: one generalised eigenproblem in a ~140 GB arena) — with per-stage
    seconds, every kernel class, and (checks) the size-independent properties of the products."""
    import tempfile

    import numpy as np
    import torch

    from driftscan_amd import beamtransfer, btgen, cylinder, device, kltransform

    log = log or (lambda *a: None)
    os.environ["DRIFTMI_STORAGE"] = "discard"
    device.reset_context()
    torch.cuda.empty_cache()
    torch.cuda.reset_peak_memory_stats()
    tel = cylinder.PolarisedCylinderTelescope.from_config(dict(CFG5))
    ctx = device.get_context(workspace_bytes=int(workspace_gb) << 30)
    rec = dict(m=int(m), nfreq=int(tel.nfreq), nbase=int(tel.nbase), lmax=int(tel.lmax), mmax=int(tel.mmax),
               sht_iter=int(tel.sht_iter))

    def sync():
        ctx.sync()
        torch.cuda.synchronize()

    with tempfile.TemporaryDirectory() as tmp:
        bt = beamtransfer.BeamTransfer(tmp, telescope=tel)
        kl = kltransform.KLTransform.from_config(dict(threshold=0.1), bt, subdir="kl")
        kl.signal(); kl.foreground()                      # host C_l tables: once per job, untimed
        # ---- BT-gen of the block
        ctx.prof_reset(2)
        sync()
        t0 = time.perf_counter()
        beam = btgen.beam_m_all(tel, ctx=ctx, max_bytes=int(bt_gb) << 30, m_range=(m, m))
        sync()
        rec["btgen_s"] = time.perf_counter() - t0
        rec["btgen_classes"] = class_table(ctx.prof_report())
        rec["beam_block_gb"] = beam.numel() * 16 / 2 ** 30
        log("configs[4] m = %d: BT-gen of the %.1f GB block %.1f s" % (m, rec["beam_block_gb"], rec["btgen_s"]))
        # ---- SVD chain + pinv, all frequencies
        ctx.prof_reset(2)
        sync()
        t0 = time.perf_counter()
        out = bt.svd_device(beam, ms=[m])     # (columns l >= m only, as generate() runs it)
        sync()
        rec["svd_s"] = time.perf_counter() - t0
        rec["svd_classes"] = class_table(ctx.prof_report())
        sv = out["singularvalues"].cpu().numpy()
        bt._dev[m] = dict(beam_svd=out["beam_svd"][0], beam_ut=out["beam_ut"][0], singularvalues=sv[0])
        svnum, _ = bt._svd_num(m)
        rec["ndof"] = int(svnum.sum())
        rec["modes_per_frequency"] = [int(svnum.min()), int(svnum.max())]
        rec["svd_sweeps"] = [int(x) for x in np.asarray(out.get("sweeps", [])).reshape(-1)][:8]
        log("configs[4] m = %d: SVD chain + pinv of %d frequencies %.1f s, ndof %d (%d..%d modes per frequency)"
            % (m, tel.nfreq, rec["svd_s"], rec["ndof"], svnum.min(), svnum.max()))
        if checks:
            T, P, L = bt.ntel, tel.num_pol_sky, tel.lmax + 1
            noisew = bt._noisew()
            wu = wp = 0.0
            for fi in (0, tel.nfreq // 2, tel.nfreq - 1):
                n = int(svnum[fi])
                if n == 0:
                    continue
                u = out["beam_ut"][0, fi, :n].cpu().numpy() / noisew[fi][None, :]
                b2 = out["beam_svd"][0, fi, :n].cpu().numpy().reshape(n, P * L)
                i2 = out["invbeam_svd"][0, fi].cpu().numpy().reshape(P * L, -1)[:, :n]
                wu = max(wu, float(np.abs(u @ u.conj().T - np.eye(n)).max()))
                wp = max(wp, float(np.abs(b2 @ i2 - np.eye(n)).max()))
            rec["check_ut_orth"], rec["check_beam_pinv"] = wu, wp
            log("configs[4] m = %d: |U U^H - I| %.2e, |beam_svd invbeam_svd - I| %.2e" % (m, wu, wp))
        del beam
        out.pop("invbeam_svd", None)
        torch.cuda.empty_cache()
        # ---- KL: covariance projections + the generalised eigenproblem of the REAL pencil
        Sh = Nh = None
        if checks:   # the pencil itself, parked in page-locked host memory (eigh_gen destroys its inputs, and the arena
            S, N, ndofs, off = kl.sn_covariance_device([m])      # of the eigensolver needs the card to itself)
            sync()
            n = int(ndofs[0])
            Sh, Nh = ctx.to_host(S[: n * n]), ctx.to_host(N[: n * n])
            del S, N
            torch.cuda.empty_cache()
        ctx.prof_reset(2)
        sync()
        t0 = time.perf_counter()
        r = kl._transform_batch([m], to_host=False)[0]
        ctx = device.get_context()
        sync()
        rec["kl_s"] = time.perf_counter() - t0
        rec["kl_classes"] = class_table(ctx.prof_report())
        ev = r[0].cpu().numpy()
        n = ev.size
        rec["kl_nkept"] = int((ev >= kl.threshold).sum())
        rec["kl_add_const"] = float(r[3]["ac"])
        rec["kl_evals_min_max"] = [float(ev.min()), float(ev.max())]
        rec["workspace_gb"] = ctx.lib.dm_ctx_workspace_bytes(ctx.h) / 2 ** 30
        log("configs[4] m = %d: KL (projections + eigh_gen, n = %d) %.1f s, %d modes kept, arena %.0f GB"
            % (m, n, rec["kl_s"], rec["kl_nkept"], rec["workspace_gb"]))
        if checks and rec["kl_nkept"] > 0:
            E = r[1]                                       # (n, n) device, rows = modes, ascending eigenvalue
            ctx.workspace_reset(1 << 30)
            torch.cuda.empty_cache()
            i0 = n - rec["kl_nkept"]
            pick = np.unique(np.linspace(i0, n - 1, min(256, rec["kl_nkept"])).astype(np.int64))
            Ek = E[torch.as_tensor(pick, device=E.device)]
            lam = torch.as_tensor(ev[pick], device=E.device)
            res_ = {}
            for name, Mh in (("N", Nh), ("S", Sh)):
                M = ctx.to_device(Mh).view(n, n)
                res_[name] = (Ek @ M) @ Ek.conj().T        # checker arithmetic (torch), not the product path
                del M
            eye = torch.eye(pick.size, dtype=res_["N"].dtype, device=E.device)
            ese = res_["S"]
            rec["check_ENE"] = float((res_["N"] - eye).abs().max().item())
            rec["check_ESE_offdiag"] = float(((ese - torch.diag(torch.diagonal(ese))).abs().max() / ese.abs().max()).item())
            rec["check_ESE_diag"] = float(((torch.diagonal(ese).real - lam).abs().max() / lam.abs().max()).item())
            log("configs[4] m = %d: |E N E^H - I| %.2e, offdiag(E S E^H)/max %.2e, diag vs lambda %.2e (sample of %d kept modes)"
                % (m, rec["check_ENE"], rec["check_ESE_offdiag"], rec["check_ESE_diag"], pick.size))
        rec["hbm_peak_gb"] = torch.cuda.max_memory_allocated() / 2 ** 30
        bt._dev.pop(m, None)
        del r, out
    beamtransfer.BeamTransfer._clcache.clear()
    device.reset_context()
    torch.cuda.empty_cache()
    # what a whole configs[4] job would cost at this block's rate (1025 blocks over 8 GPUs, BT-gen in calls of two blocks)
    per_block = rec["svd_s"] + rec["kl_s"] + rec["btgen_s"]
    rec["per_block_s"] = per_block
    rec["projected_8gpu_job_h"] = per_block * (tel.mmax + 1) / 8.0 / 3600.0
    rec["projection_note"] = ("%d m-blocks / 8 GPUs x (BT-gen + SVD + KL of this block); m = %d has about the median ndof — the low-m "
                              "blocks cost more in KL, the high-m ones less (ndof falls with m)" % (tel.mmax + 1, m))
    return rec
