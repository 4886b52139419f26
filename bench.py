#!/usr/bin/env python3
"""bench.py — m-blocks/sec of the per-m hot path (BT-gen + SVD + KL) on MI355X.

Workload (BASELINE.json configs[1]): 32-feed unpolarised cylinder
(2 cylinders x 16 feeds, width 5 m, spacing 0.4 m), 16 channels 400-450 MHz (edge),
force_lmax = force_mmax = 128  =>  nbase 46, ntel 92, 129 m-blocks, ndofmax 1472.
One "step" = one pass of the whole hot path over all 129 m-blocks with the telescope
description resident (beam-transfer generation -> three-stage SVD compression + pinv ->
covariance projection + generalised eigenproblem of the KL transform); products stay in
HBM, file output is not part of the timed region.  With N > 1 ranks every GPU runs the
same 129-block workload (independent m-blocks, no data-path collective): weak scaling,
value = N * 129 * steps / max-over-ranks time.

Prints ONE JSON line on rank 0 (see the README of the driver contract).
"""
import argparse
import json
import os
import sys
import time

# one process per GPU: keep the host BLAS / OpenMP pools of the ranks from oversubscribing the node
if int(os.environ.get("WORLD_SIZE", "1")) > 1:
    os.environ.setdefault("OMP_NUM_THREADS", str(max(1, (os.cpu_count() or 8) // int(os.environ["WORLD_SIZE"]))))

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

CFG2 = dict(num_freq=16, freq_start=400.0, freq_end=450.0, freq_mode="edge", num_cylinders=2, cylinder_width=5.0,
            num_feeds=16, feed_spacing=0.4, tsys=1.0, force_lmax=128, force_mmax=128)
FP64_MFMA_PEAK_TFLOPS = 78.6  # MI355X fp64 matrix peak (AMD datasheet)
HBM_PEAK_GBS = 8000.0         # HBM3E spec (MI355X_MICROARCH.md: 8 TB/s peak, ~6.3 achievable)
HBM_CLASSES = ("trd_symv", "trd_wx")


def build_objects(tmpdir):
    from driftscan_amd import beamtransfer, cylinder, kltransform

    tel = cylinder.UnpolarisedCylinderTelescope.from_config(CFG2)
    bt = beamtransfer.BeamTransfer(tmpdir, telescope=tel)
    kl = kltransform.KLTransform.from_config(dict(threshold=0.1), bt, subdir="kl")
    return tel, bt, kl


_pool = None


def _svd_kl_group(bt, kl, beam_all, ms):
    """SVD chain + KL for one group of m-blocks on the calling thread's context; returns
    (seconds in SVD, seconds in KL) as seen by this thread."""
    import torch

    from driftscan_amd import device

    ctx = device.get_context()
    t0 = time.perf_counter()
    idx = torch.as_tensor(ms, device=beam_all.device)
    res = bt.svd_device(beam_all.index_select(0, idx))                # SVD chain + pinv, the whole group at once
    sv = res["singularvalues"].cpu().numpy()
    ctx.sync()
    t1 = time.perf_counter()
    for i, mi in enumerate(ms):
        bt._dev[mi] = dict(beam_svd=res["beam_svd"][i], beam_ut=res["beam_ut"][i], singularvalues=sv[i])
    out = None
    for batch in kl._batches(list(ms)):
        out = kl._transform_batch(batch, to_host=False)               # projections + eigh_gen, products stay in HBM
    ctx.sync()
    return t1 - t0, time.perf_counter() - t1, out


def hot_path_step(tel, bt, kl, ctx, stage_times=None, streams=1):
    """One pass over all m-blocks; everything stays on the device.  After the beam-transfer
    generation the m-blocks are dealt round-robin into `streams` groups, each driven by its own
    thread / libdriftmi context / HIP stream, so that the launch-latency-bound phases of one
    group (Jacobi sweeps, the tail of the tridiagonalisation) overlap with the MFMA- and
    HBM-bound phases of the other."""
    import torch

    from driftscan_amd import btgen

    global _pool
    t0 = time.perf_counter()
    beam_all = btgen.beam_m_all(tel, ctx=ctx)                         # (mmax+1, F, 2, B, P, L)
    ctx.sync()
    torch.cuda.synchronize()
    t1 = time.perf_counter()
    ms = list(range(tel.mmax + 1))
    groups = [ms[g::streams] for g in range(streams)]
    if streams == 1:
        parts = [_svd_kl_group(bt, kl, beam_all, groups[0])]
    else:
        if _pool is None or _pool._max_workers != streams:
            from concurrent.futures import ThreadPoolExecutor

            _pool = ThreadPoolExecutor(max_workers=streams)
        futs = [_pool.submit(_svd_kl_group, bt, kl, beam_all, g) for g in groups]
        parts = [f.result() for f in futs]
    torch.cuda.synchronize()
    t3 = time.perf_counter()
    if stage_times is not None:
        tsvd = max(p[0] for p in parts)
        stage_times.append((t1 - t0, tsvd, (t3 - t1) - tsvd))
    return parts[0][2]


def cpu_baseline(tel, bt, kl, budget_s=25.0):
    """The oracle (numpy/scipy restatement, kind = "port") timed on the host cores for a
    bounded sample of m-blocks of the same workload: SVD chain + KL for m in a spread of
    values, plus BT-gen for a handful of (f, b) columns scaled to the full count."""
    import scipy

    from oracle import btgen as ob
    from oracle import kl as okl
    from oracle import svdchain as osvd

    ncores = os.cpu_count() or 1
    desc = dict(polarised=False, zenith=tel.zenith, baselines=tel.baselines, uniquepairs=tel.uniquepairs,
                beamclass=tel.beamclass, wavelengths=tel.wavelengths, cylinder_width=tel.cylinder_width,
                fwhm_e=tel.fwhm_e, fwhm_h=tel.fwhm_h, lmax=tel.lmax, mmax=tel.mmax, l_boost=tel.l_boost,
                included_freq=np.array([0]), included_baseline=np.array([0, tel.nbase - 1]),
                accuracy_boost=tel.accuracy_boost)
    t0 = time.perf_counter()
    ob.beam_transfer_m(desc)  # 2 of the F*B columns, all m
    t_bt_cols = time.perf_counter() - t0
    t_bt_full = t_bt_cols / 2.0 * tel.nfreq * tel.nbase
    # SVD + KL on synthetic blocks of the right shape (random full-rank, like SURVEY §6's probe)
    rng = np.random.default_rng(1000)
    F, B, L = tel.nfreq, tel.nbase, tel.lmax + 1
    noisew = bt._noisew()[:, :B]
    npw = kl._npower(1.0)
    sample_m = [0, tel.mmax // 4, tel.mmax // 2, 3 * tel.mmax // 4]
    t_svdkl = 0.0
    done = 0
    for mi in sample_m:
        blk = np.zeros((F, 2, B, 1, L), dtype=np.complex128)
        blk[..., mi:] = (rng.standard_normal((F, 2, B, 1, L - mi)) + 1j * rng.standard_normal((F, 2, B, 1, L - mi))) \
            * np.exp(-np.arange(L - mi) / 20.0)
        t0 = time.perf_counter()
        o = osvd.svd_m(blk, noisew, polsvcut=bt.polsvcut)
        cs, cn = okl.sn_covariance(o["beam_svd"], o["beam_ut"], o["singularvalues"], kl.signal(), kl.foreground(),
                                   npw, svcut=bt.svcut)
        okl.kl_transform_m(cs, cn)
        t_svdkl += time.perf_counter() - t0
        done += 1
        if t_svdkl + t_bt_cols > budget_s:
            break
    per_m = t_svdkl / done + t_bt_full / (tel.mmax + 1)
    return dict(value=1.0 / per_m, unit="m-blocks/s", cores=ncores, kind="port",
                sample="oracle (numpy %s / scipy %s, threaded BLAS on %d cores): SVD chain + KL on %d m-blocks "
                       "(m = %s) of the config-2 shape, BT-gen on 2 of %d (f,b) columns scaled to all"
                       % (np.__version__, scipy.__version__, ncores, done, sample_m[:done], tel.nfreq * tel.nbase))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=3)
    ap.add_argument("--warmup", type=int, default=1)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--all-modes", action="store_true",
                    help="form every KL mode (subset = False) instead of only the ones transform_save keeps")
    ap.add_argument("--streams", type=int, default=int(os.environ.get("DRIFT_BENCH_STREAMS", "1")),
                    help="concurrent m-block groups per GPU (threads x HIP streams); 2 gives +7 %% m-blocks/s but the "
                         "per-kernel durations (and so the roofline figure) then include the interference")
    args = ap.parse_args()

    import tempfile

    import torch

    from driftscan_amd import device, parallel

    world = int(os.environ.get("WORLD_SIZE", "1"))
    force_dist = os.environ.get("DRIFT_BENCH_FORCE_DIST") == "1"  # exercise the RCCL path on one GPU
    if world > 1 or force_dist:
        import torch.distributed as dist

        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29511")
        os.environ.setdefault("RANK", "0")
        os.environ.setdefault("WORLD_SIZE", "1")
        torch.cuda.set_device(int(os.environ.get("LOCAL_RANK", "0")))
        dist.init_process_group(backend="nccl")
    rank = parallel.rank()
    local = int(os.environ.get("LOCAL_RANK", "0"))
    torch.cuda.set_device(local)
    ctx = device.get_context(workspace_bytes=24 << 30)

    with tempfile.TemporaryDirectory() as tmp:
        tel, bt, kl = build_objects(tmp)
        if args.all_modes:
            kl.subset = False
        for _ in range(args.warmup):
            hot_path_step(tel, bt, kl, ctx, streams=args.streams)
        # A full (generation-2) cycle collection walks every object torch/numpy created at import
        # (~45 ms here) and would land at a random point of the timed region: collect now and move
        # the survivors to the permanent generation, as a long-running pipeline process would.
        import gc

        gc.collect()
        gc.freeze()
        for c in list(device._all):  # every context (main thread + stream workers)
            c.prof_reset(True)
        stage = []
        parallel.barrier()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(args.steps):
            hot_path_step(tel, bt, kl, ctx, stage_times=stage, streams=args.streams)
        torch.cuda.synchronize()
        parallel.barrier()
        dt = time.perf_counter() - t0
        prof = {}
        for c in list(device._all):
            for k, v in c.prof_report().items():
                a = prof.setdefault(k, dict(ms=0.0, flops=0.0, launches=0))
                a["ms"] += v["ms"]; a["flops"] += v["flops"]; a["launches"] += v["launches"]
        if world > 1 or force_dist:
            import torch.distributed as dist

            tmax = torch.tensor([dt], dtype=torch.float64, device="cuda")
            dist.all_reduce(tmax, op=dist.ReduceOp.MAX)
            dt = float(tmax.item())
        nblocks = tel.mmax + 1
        value = world * nblocks * args.steps / dt
        if rank == 0:
            st = np.array(stage).mean(axis=0)
            dom = max(prof, key=lambda k: prof[k]["ms"] * (8.0 if k in HBM_CLASSES else 1.0)) if prof else None
            roofline = None
            # HBM bytes per launch of the dominant kernel: PMC counters cannot be read from inside this
            # process, so the figure comes from the committed rocprofv3 --pmc passes of this same
            # command (profiles/*_pmc_traffic.json, made by scratch/run_profiles.sh + make_traffic_json.py).
            traffic, traffic_src = None, None
            try:
                import glob

                tj = sorted(glob.glob(os.path.join(ROOT, "profiles", "*_pmc_traffic.json")))
                if tj and dom is not None:
                    rec = json.load(open(tj[-1]))
                    key = {"zgemm_grouped": "zgemm_grouped_kernel<false, false>", "trd_symv": "trd_symv_kernel",
                           "trd_wx": "trd_wx_kernel", "gemm_grouped_realB": "zgemm_grouped_kernel<true, false>",
                           "dgemm_grouped": "dgemm_grouped_kernel", "jac_inner": "jac_inner_kernel<false>",
                           "jac_gram": "jac_gram_kernel", "jac_apply": "jac_apply_kernel"}.get(dom)
                    if key is not None and key not in rec:  # kernels compiled inside a namespace (dm_trd32::trd_symv_kernel)
                        key = next((k for k in rec if k.endswith("::" + key)), key)
                    if key in rec:
                        traffic = rec[key]["fetch_bytes_per_launch"] + rec[key]["write_bytes_per_launch"]
                        traffic_src = os.path.relpath(tj[-1], ROOT)
            except Exception:
                traffic, traffic_src = None, None
            if dom is not None:
                p = prof[dom]
                secs = p["ms"] * 1e-3
                if dom in HBM_CLASSES:  # work counter = algorithmic bytes
                    ach = p["flops"] / secs / 1e9 if secs > 0 else 0.0
                    roofline = dict(bound="hbm", kernel=dom, achieved=ach, peak=HBM_PEAK_GBS, unit="GB/s",
                                    frac=ach / HBM_PEAK_GBS, traffic=traffic, traffic_unit="bytes/launch",
                                    traffic_source=traffic_src, launches=p["launches"],
                                    avg_launch_us=1e3 * p["ms"] / max(p["launches"], 1),
                                    bytes_per_launch=p["flops"] / max(p["launches"], 1),
                                    sampling="every 8th launch of the column loop is timed (uniform in k); "
                                             "`launches` counts the timed ones")
                else:
                    ach = p["flops"] / secs / 1e12 if secs > 0 else 0.0
                    roofline = dict(bound="mfma", kernel=dom, achieved=ach, peak=FP64_MFMA_PEAK_TFLOPS, unit="TFLOP/s",
                                    frac=ach / FP64_MFMA_PEAK_TFLOPS, traffic=traffic, traffic_unit="bytes/launch",
                                    traffic_source=traffic_src, launches=p["launches"],
                                    avg_launch_us=1e3 * p["ms"] / max(p["launches"], 1),
                                    flops_per_launch=p["flops"] / max(p["launches"], 1))
                # the runner-up class of the other kind, for context (MFMA vs HBM side of the step)
                others = [k for k in prof if (k in HBM_CLASSES) != (dom in HBM_CLASSES)]
                if others:
                    o = max(others, key=lambda k: prof[k]["ms"] * (8.0 if k in HBM_CLASSES else 1.0))
                    q = prof[o]
                    if o in HBM_CLASSES:
                        oa = q["flops"] / (q["ms"] * 1e-3) / 1e9
                        roofline["also"] = dict(kernel=o, bound="hbm", achieved=oa, unit="GB/s", frac=oa / HBM_PEAK_GBS,
                                                ms_per_step=8.0 * q["ms"] / args.steps)
                    else:
                        oa = q["flops"] / (q["ms"] * 1e-3) / 1e12
                        roofline["also"] = dict(kernel=o, bound="mfma", achieved=oa, unit="TFLOP/s",
                                                frac=oa / FP64_MFMA_PEAK_TFLOPS, ms_per_step=q["ms"] / args.steps)
            line = {
                "metric": "m-blocks/sec (BT-gen + SVD + KL)",
                "value": value,
                "unit": "m-blocks/s",
                "n_gpus": world,
                "steps": args.steps,
                "warmup": args.warmup,
                "ms_per_step": 1e3 * dt / args.steps,
                "higher_is_better": True,
                "scaling": "weak",
                "vs_baseline": None,
                "dtype": "f64",
                "data": "synthetic",
                "config": {"workload": "configs[1]: 32-feed unpolarised cylinder, nfreq=16, nbase=46, lmax=mmax=128, "
                                       "129 m-blocks per GPU per step, KLTransform with foregrounds",
                           "nfreq": 16, "nbase": 46, "lmax": 128, "mmax": 128, "sharding": "m-blocks, replicas per GPU", "streams_per_gpu": args.streams,
                           "kl_products": "all eigenvalues + every mode" if args.all_modes else
                           "all eigenvalues + the modes with S/N >= threshold (subset = True, what transform_save writes)"},
                "stage_ms": {"btgen": 1e3 * st[0], "svd": 1e3 * st[1], "kl": 1e3 * st[2]},
                # the two tridiagonalisation classes are timed on every 8th launch: scaled back to all launches
                "kernels_ms": {k: v["ms"] / args.steps * (8.0 if k in HBM_CLASSES else 1.0) for k, v in prof.items()},
                "roofline": roofline,
                "cpu_baseline": None if args.no_cpu_baseline else cpu_baseline(tel, bt, kl),
            }
            print(json.dumps(line))
    if world > 1 or force_dist:
        import torch.distributed as dist

        dist.destroy_process_group()


if __name__ == "__main__":
    main()
