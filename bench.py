#!/usr/bin/env python3
"""bench.py — m-blocks/sec of the per-m hot path (BT-gen + SVD + KL) on MI355X.

Workload (BASELINE.json configs[1]): 32-feed unpolarised cylinder
(2 cylinders x 16 feeds, width 5 m, spacing 0.4 m), 16 channels 400-450 MHz (edge),
force_lmax = force_mmax = 128  =>  nbase 46, ntel 92, 129 m-blocks, ndofmax 1472.
One "step" = one pass of the whole hot path over the m-blocks of the rank with the telescope
description resident (beam-transfer generation -> three-stage SVD compression + pinv ->
covariance projection + generalised eigenproblem of the KL transform); products stay in
HBM, file output is not part of the timed region.

    python bench.py --gpus N --steps K --warmup W [--mode weak|sharded]

`--gpus N` with N > 1 and no launcher environment (WORLD_SIZE unset): this process starts N
rank processes itself — before it makes any GPU call — one per GPU over RCCL ("nccl"), relays
rank 0's JSON line and exits non-zero if a rank fails.  Under `torch.distributed.run` the
RANK/LOCAL_RANK/WORLD_SIZE of the launcher are used.

Modes (m-blocks are independent: no data-path collective in either; weak is the default at every N):
  weak     every rank runs the full 129-block workload (per-GPU work fixed); the singular-value and
           eigenvalue spectra of every rank are gathered to rank 0 inside the timed region, as
           `_collect_svd_spectrum` / `KLTransform._collect` do.  value = N * 129 * steps / time.
  sharded  ONE 129-block workload split over the ranks in contiguous m-ranges of equal estimated
           cost (`parallel.partition_contiguous`); each rank generates, compresses and transforms only
           its own blocks; the spectra gather and an all-reduce of a per-rank (nbands x nbands)
           matrix (the Fisher assembly pattern, psestimation.py:506-507) are inside the timed
           region.  value = 129 * steps / time ("scaling": "strong").

Output (rank 0): the full record as `bench_detail {...}` on an EARLIER stdout line and in `bench_detail.json`
($DRIFT_BENCH_DETAIL), then — LAST — the compact driver line (< 8 KB; `benchlib/line.py`): the contract keys, `config`, `roofline`
(dominant kernel class: algorithmic flops per launch / its average launch duration by HIP events over the timed region, peak,
frac, PMC traffic per launch), `cpu_baseline` (the oracle on the host cores, N = 1 only), `parity`, `stage_ms`, per-rank step
times, `north_star` (live shares of the configs[2] job; at N > 1 the real N-rank job).

The legs live in `benchlib/`: `common` (telescopes, peaks, build id), `cpu` (CPU baseline + parity), `jobs` (shares and jobs
through `ProductManager.generate()`), `northstar` (the north-star leg of the default line), `line` (the compact line).
"""
import argparse
import json
import os
import subprocess
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

from benchlib.common import (CFG2, CFG3, CFG5, EXT_CLASSES, FP64_MFMA_PEAK_TFLOPS, HBM_CLASSES, HBM_PEAK_GBS, VALU_CLASSES,  # noqa: E402,F401
                             build_id, class_table, host_cores, stage_work)
from benchlib.cpu import cpu_baseline, cpu_worker_main  # noqa: E402,F401
from benchlib.jobs import job_budgets, job_conf, measure_configs4_block, measure_job, measure_share  # noqa: E402,F401
from benchlib.northstar import north_star_job_leg, north_star_leg  # noqa: E402,F401


def parse_args(argv=None):
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=5)
    ap.add_argument("--warmup", type=int, default=2)
    ap.add_argument("--prime-passes", type=int, default=10,
                    help="untimed passes of the hot path before the warm-up steps (settles a fresh box; 0 to skip)")
    ap.add_argument("--mode", choices=["weak", "sharded"], default=None,
                    help="default: weak (a full 129-block workload per GPU); sharded: ONE workload in cost-balanced m-ranges over the ranks")
    ap.add_argument("--workload", choices=["configs1", "configs2", "configs3", "configs4"], default="configs1",
                    help="configs1 = BASELINE configs[1] (the default line); configs2 / configs3 = the north-star job "
                         "(128-feed polarised cylinder, nfreq 64, lmax 512; configs3 adds DoubleKL + the exact Fisher matrix) "
                         "through ProductManager.generate(): one rank's share of the 8-GPU job on this GPU")
    ap.add_argument("--share", default="0/8", help="--workload configs2|configs3: which rank's share, as r/N")
    ap.add_argument("--job", action="store_true",
                    help="--workload configs2|configs3: the WHOLE job on --gpus N real ranks (one process per GPU over --backend), "
                         "instead of one emulated share")
    ap.add_argument("--m", type=int, default=300, help="--workload configs4: which m-block")
    ap.add_argument("--no-checks", action="store_true", help="--workload configs4: skip the property checks of the products")
    ap.add_argument("--files", action="store_true",
                    help="--workload configs2|configs3: write the product files (default: products stay in HBM)")
    ap.add_argument("--truncate", action="store_true",
                    help="--workload configs2|configs3: truncate the beam transfer blocks on the device (the reference's truncate=True)")
    ap.add_argument("--outdir", default=None, help="--files: where the temporary product directory is made (default: TMPDIR)")
    ap.add_argument("--share-mmax", type=int, default=None, help=argparse.SUPPRESS)  # any value: toy telescope (rehearsal)
    ap.add_argument("--backend", choices=["nccl", "gloo"], default="nccl",
                    help="gloo + --one-gpu rehearses the multi-rank path on a single card (RCCL refuses two ranks per GPU)")
    ap.add_argument("--one-gpu", action="store_true", help="every rank uses cuda:0 (rehearsal on a one-GPU box)")
    ap.add_argument("--shard", default=None, metavar="r/N",
                    help="configs1, one process: run ONLY the m-range rank r of N would own in --mode sharded (no process group) — "
                         "the per-rank step time behind the expected strong-scaling curve (DESIGN.md section 6)")
    ap.add_argument("--no-rebalance", action="store_true",
                    help="--mode sharded: keep the m-ranges of the static cost model instead of rebalancing them on measured times")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-north-star", action="store_true",
                    help="skip the north-star leg of the default line (one rank's share of the configs[2] job through "
                         "ProductManager.generate() after the timed configs[1] region; rank 0 at --gpus 1 only)")
    ap.add_argument("--north-star-share", default=os.environ.get("DRIFT_BENCH_NS_SHARE", "0/8"),
                    help="which share the north-star leg runs (r/N of the configs[2] job)")
    ap.add_argument("--all-modes", action="store_true",
                    help="form every KL mode (subset = False) instead of only the ones transform_save keeps")
    ap.add_argument("--streams", type=int, default=int(os.environ.get("DRIFT_BENCH_STREAMS", "1")),
                    help="concurrent m-block groups per GPU (threads x HIP streams); 2 gave +7 %% m-blocks/s in round 2, "
                         "and is slower since the two-stage tridiagonalisation (937 against 1020: the persistent chase kernels of "
                         "the groups compete for the same CUs)")
    ap.add_argument("--cpu-worker", nargs=2, metavar=("JOBS", "NPROC"), help=argparse.SUPPRESS)
    return ap.parse_args(argv)


# ---------------------------------------------------------------------------------------------------
# launcher: N rank processes from a parent that never touches the GPU
# ---------------------------------------------------------------------------------------------------
def launch_ranks(args, argv):
    import socket

    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    import tempfile

    procs = []
    out0f = tempfile.TemporaryFile()   # rank 0's stdout (detail line + driver line): a file, not a pipe nobody drains
    for r in range(args.gpus):
        env = dict(os.environ)
        env.update(RANK=str(r), LOCAL_RANK=str(r), WORLD_SIZE=str(args.gpus), MASTER_ADDR="127.0.0.1",
                   MASTER_PORT=str(port), HSA_ENABLE_IPC_MODE_LEGACY="0", DRIFT_BENCH_CHILD="1")
        procs.append(subprocess.Popen([sys.executable, os.path.abspath(__file__)] + argv, env=env,
                                      stdout=out0f if r == 0 else subprocess.DEVNULL))
    rc = 0
    try:
        while True:
            states = [p.poll() for p in procs]
            bad = [st for st in states if st not in (None, 0)]
            if bad:  # a rank failed: the others would wait for it in a collective until the timeout
                rc = bad[0]
                break
            if all(st == 0 for st in states):
                break
            time.sleep(0.2)
    finally:
        for p in procs:
            if p.poll() is None:
                p.kill()   # exactly the PIDs started above
        for p in procs:
            p.wait()
        out0f.seek(0)
        out0 = out0f.read() or b""
        out0f.close()
    sys.stdout.write(out0.decode())
    sys.stdout.flush()
    return rc


# ---------------------------------------------------------------------------------------------------
# the hot path
# ---------------------------------------------------------------------------------------------------
def build_objects(tmpdir):
    from driftscan_amd import beamtransfer, cylinder, kltransform

    tel = cylinder.UnpolarisedCylinderTelescope.from_config(CFG2)
    bt = beamtransfer.BeamTransfer(tmpdir, telescope=tel)
    kl = kltransform.KLTransform.from_config(dict(threshold=0.1), bt, subdir="kl")
    return tel, bt, kl


_pool = None
_BUSY = [0.0]   # seconds this rank spent in its own BT-gen + SVD + KL (without the waits inside the collectives)
_NKEEP = {}   # m -> eigenvectors actually back-transformed in the last pass (the modes transform_save keeps)


def _svd_kl_group(bt, kl, beam_all, ms, m0=0, ready=None):
    """SVD chain + KL for one group of m-blocks on the calling thread's context; returns
    (seconds in SVD, seconds in KL, KL products, singular values (host), ndofs)."""
    import torch

    from driftscan_amd import device

    ctx = device.get_context()
    if ready is not None:
        device.wait_for(ready)       # the blocks come from the main thread's stream
    t0 = time.perf_counter()
    if len(ms) and list(ms) == list(range(ms[0], ms[-1] + 1)):
        blocks = beam_all[ms[0] - m0 : ms[-1] - m0 + 1]                 # a contiguous range of m: a view (what generate() takes)
    else:   # groups dealt round-robin (--streams > 1): gathered — the index upload waits for the stream, the copy moves the blocks
        blocks = beam_all.index_select(0, torch.as_tensor([m - m0 for m in ms], device=beam_all.device))
    res = bt.svd_device(blocks, ms=list(ms))                           # SVD chain + pinv, the whole group at once
    sv = res["singularvalues"].cpu().numpy()
    ctx.sync()
    t1 = time.perf_counter()
    views = bt._register_sv(ms, sv)
    bsvd, but = res["beam_svd"].unbind(0), res["beam_ut"].unbind(0)    # all per-m views in one call each
    for i, mi in enumerate(ms):
        bt._dev[mi] = dict(beam_svd=bsvd[i], beam_ut=but[i], singularvalues=views[i])
    out = []
    nkeep = {}
    for batch in kl._batches(list(ms)):
        out += kl._transform_batch(batch, to_host=False)             # projections + eigh_gen, products stay in HBM
        lk = getattr(ctx, "last_nkeep", None)
        if lk is not None and len(lk) == len(batch):
            nkeep.update({mi: int(k) for mi, k in zip(batch, lk)})
    ctx.sync()
    _NKEEP.update(nkeep)
    return t1 - t0, time.perf_counter() - t1, out, sv


def hot_path_step(tel, bt, kl, ctx, stage_times=None, streams=1, m_range=None, collect=False, keep=None, spectra=None):
    """One pass over the m-blocks of this rank (all of them, or the contiguous `m_range`); everything stays on
    the device.  After the beam-transfer generation the m-blocks are dealt round-robin into `streams` groups,
    each driven by its own thread / libdriftmi context / HIP stream.  With `collect` the spectra of all ranks are
    gathered to rank 0 (beamtransfer.py:931-947, kltransform.py:452-478) and a per-rank band matrix is all-reduced
    (the pattern of the Fisher assembly, psestimation.py:506-507)."""
    import numpy as np
    import torch

    from driftscan_amd import btgen, parallel

    global _pool
    t0 = time.perf_counter()
    beam_all = btgen.beam_m_all(tel, ctx=ctx, m_range=m_range)        # (nm, F, 2, B, P, L)
    if stage_times is not None:   # only the pass that reports per-stage times waits here: BT-gen returns when queued, and
        ctx.sync()                # the host prepares the SVD stage while the GPU still transforms (as generate() does)
        torch.cuda.synchronize()
    t1 = time.perf_counter()
    m0 = 0 if m_range is None else m_range[0]
    ms = list(range(tel.mmax + 1)) if m_range is None else list(range(m_range[0], m_range[1] + 1))
    if streams > 1 and os.environ.get("DRIFT_BENCH_SPLIT") == "contig":
        groups = [parallel.partition_contiguous(ms, [bt._m_cost(m) for m in ms], n=streams, r=g) for g in range(streams)]
    else:
        groups = [ms[g::streams] for g in range(streams)]
    if streams == 1:
        parts = [_svd_kl_group(bt, kl, beam_all, groups[0], m0)]
    else:
        os.environ["DRIFTMI_THREAD_STREAMS"] = "1"   # each group's thread on a HIP stream of its own (device.get_context)
        if _pool is None or _pool._max_workers != streams:
            from concurrent.futures import ThreadPoolExecutor

            _pool = ThreadPoolExecutor(max_workers=streams)
        ready = ctx.record_event()
        futs = [_pool.submit(_svd_kl_group, bt, kl, beam_all, g, m0, ready) for g in groups]
        parts = [f.result() for f in futs]
    torch.cuda.synchronize()
    t3 = time.perf_counter()
    _BUSY[0] += t3 - t0
    tcoll = 0.0
    if collect:
        mine = []
        for g, p in zip(groups, parts):
            flat = torch.cat([r[0].reshape(-1) for r in p[2]]).cpu().numpy()   # one copy for all spectra of the group
            cuts = np.cumsum([0] + [int(r[0].numel()) for r in p[2]])
            mine += [(mi, p[3][i], flat[cuts[i] : cuts[i + 1]]) for i, mi in enumerate(g)]
        allparts = parallel.gather_objects(mine)
        band = np.zeros((9, 9))
        for _, sv, ev in mine:
            band[0, 0] += float(ev.sum())
        parallel.allreduce_sum(band)
        if parallel.rank0():
            assert sum(len(p) for p in allparts) >= len(mine)
        tcoll = time.perf_counter() - t3
    if keep is not None:
        for mi in keep:
            if m0 <= mi < m0 + beam_all.shape[0]:
                keep[mi] = beam_all[mi - m0].cpu().numpy()
    if spectra is not None:   # the spectra of this pass on the host: singular values (F, K) and all eigenvalues per m
        for g, p in zip(groups, parts):
            for i, mi in enumerate(g):
                spectra.setdefault("sv", {})[mi] = np.asarray(p[3][i])
                spectra.setdefault("ev", {})[mi] = p[2][i][0].cpu().numpy()
    if stage_times is not None:
        tsvd = max(p[0] for p in parts)
        stage_times.append((t1 - t0, tsvd, (t3 - t1) - tsvd, tcoll))
    return parts[0][2]


def run_job(args):
    """`bench.py --workload configs2|configs3 --job [--gpus N]`: under a launcher (WORLD_SIZE set) this process is one rank
    of the job; without one it starts the N ranks itself (before touching the GPU)."""
    if "WORLD_SIZE" not in os.environ:
        return launch_ranks(args, sys.argv[1:])
    line = measure_job(args.workload, backend=args.backend, one_gpu=args.one_gpu, toy=bool(args.share_mmax))
    if line is not None:
        print(json.dumps(line))
        sys.stdout.flush()
    return 0


def run_share(args):
    line = measure_share(args.workload, args.share, files=args.files, share_mmax=args.share_mmax, truncate=args.truncate,
                         outdir=args.outdir)
    print(json.dumps(line))
    sys.stdout.flush()
    return 0


def run_configs4(args):
    rec = measure_configs4_block(args.m, checks=not args.no_checks,
                                 log=lambda *a: print(time.strftime("%H:%M:%S"), *a, file=sys.stderr, flush=True))
    line = {"metric": "m-blocks/sec (BT-gen + SVD + KL)", "value": 1.0 / rec["per_block_s"], "unit": "m-blocks/s", "n_gpus": 1,
            "steps": 1, "warmup": 0, "ms_per_step": 1e3 * rec["per_block_s"], "higher_is_better": True, "scaling": "weak",
            "vs_baseline": None, "dtype": "f64", "data": "synthetic",
            "config": {"workload": "configs[4]: CHIME-like 512-feed polarised cylinder, nfreq=256, nbase=1776, lmax=mmax=1024: ONE "
                                   "real m-block (m = %d) through BeamTransfer / KLTransform" % rec["m"],
                       "nfreq": 256, "nbase": 1776, "lmax": 1024, "mmax": 1024, "sht_iter": rec["sht_iter"]},
            "block": rec, "roofline": None, "cpu_baseline": None}
    print(json.dumps(line))
    sys.stdout.flush()
    return 0


def main():
    args = parse_args()
    if args.cpu_worker:
        cpu_worker_main(*args.cpu_worker)
        return 0
    if args.workload == "configs4":
        return run_configs4(args)
    if args.workload != "configs1":
        return run_job(args) if args.job else run_share(args)
    if args.mode is None:
        # m-blocks are independent units with no data-path collective: the units are sharded over the ranks at a FIXED number
        # per GPU — every rank takes a full 129-block workload (as every rank of a large job takes its share of the blocks), the
        # spectra of all ranks are gathered inside the timed region: "scaling": "weak", value = all ranks' blocks / time.
        # `--mode sharded` splits ONE 129-block workload instead (strong scaling: a lock-step chain whose length follows the
        # largest matrix of a range, 1.2 x / 1.8 x / 2.3 x expected at 2 / 4 / 8 ranks); the north-star
        # leg of the N > 1 line runs the REAL m-sharded configs[2] job on the N ranks either way.
        args.mode = "weak"
    launched = "WORLD_SIZE" in os.environ
    if args.gpus > 1 and not launched:
        # no launcher: start the ranks ourselves, before anything here touches the GPU
        return launch_ranks(args, sys.argv[1:])
    world = int(os.environ.get("WORLD_SIZE", "1"))
    # one process per GPU: keep the host BLAS / OpenMP pools of the ranks from oversubscribing the node
    if world > 1:
        os.environ.setdefault("OMP_NUM_THREADS", str(max(1, (os.cpu_count() or 8) // world)))

    import tempfile

    import numpy as np
    import torch

    from driftscan_amd import device, parallel

    force_dist = os.environ.get("DRIFT_BENCH_FORCE_DIST") == "1"  # exercise the collective path with one rank
    local = 0 if args.one_gpu else int(os.environ.get("LOCAL_RANK", "0"))
    if args.one_gpu:
        os.environ["DRIFTMI_DEVICE"] = "0"
    if world > 1 or force_dist:
        import torch.distributed as dist

        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29511")
        os.environ.setdefault("RANK", "0")
        os.environ.setdefault("WORLD_SIZE", "1")
        torch.cuda.set_device(local)
        dist.init_process_group(backend=args.backend)
    rank = parallel.rank()
    torch.cuda.set_device(local)
    ctx = device.get_context(workspace_bytes=24 << 30)

    with tempfile.TemporaryDirectory() as tmp:
        tel, bt, kl = build_objects(tmp)
        if args.all_modes:
            kl.subset = False
        nblocks = tel.mmax + 1
        m_range = None
        if args.mode == "sharded" and world > 1:
            # the product's own cost model (BeamTransfer._m_cost: BT-gen flat in m, the SVD chain linear, the KL stage cubic
            # in the number of l >= m) — the ranges `ProductManager.generate()` would give the ranks
            allm = list(range(nblocks))
            mine = parallel.partition_contiguous(allm, [bt._m_cost(m) for m in allm])
            m_range = (mine[0], mine[-1])
        if args.shard:
            sr, sn = (int(x) for x in args.shard.split("/"))
            allm = list(range(nblocks))
            mine = parallel.partition_contiguous(allm, [bt._m_cost(m) for m in allm], n=sn, r=sr)
            m_range = (mine[0], mine[-1])
            args.no_cpu_baseline = args.no_north_star = True
        collect = world > 1 or force_dist
        # Device priming, before the W warm-up steps: a fresh box runs its first ~1.5 s of GPU work about 4 % slower
        # (clocks and page mappings settle; measured: first process of a box 719 m-blocks/s at W = 2, 756 at W = 12,
        # every later process 753 at W = 2).  The same hot path, untimed, for PRIME_S seconds of wall time; every rank
        # runs the same number of passes so that the collectives inside them match.
        t_prime = time.perf_counter()
        for _ in range(args.prime_passes):
            hot_path_step(tel, bt, kl, ctx, streams=args.streams, m_range=m_range, collect=collect)
        torch.cuda.synchronize()
        t_prime = time.perf_counter() - t_prime
        rebalance_log = None
        if args.mode == "sharded" and world > 1 and not args.no_rebalance:
            # MEASURED load balance (untimed, before the warm-up): the static cost model of `_m_cost` is calibrated on the
            # configs[2] job; at this small workload a rank's step is a lock-step chain whose length follows the largest
            # matrix of its range, and the low-m ranks come out up to 1.6 x slower than the high-m ones.  A few rounds of:
            # every rank times two passes over its range, the times are all-gathered, the cost of an m is taken as its
            # rank's time / its rank's blocks, and the contiguous partition is recomputed from those costs (every rank
            # computes the same boundaries); the best partition seen is kept.
            import torch.distributed as dist

            dev_ = "cuda" if args.backend == "nccl" else "cpu"
            allm = list(range(nblocks))
            ranges = None
            best = (float("inf"), None)
            rebalance_log = []
            for rnd in range(4):
                _BUSY[0] = 0.0
                for _ in range(2):
                    hot_path_step(tel, bt, kl, ctx, streams=args.streams, m_range=m_range, collect=collect)
                mine_t = torch.tensor([_BUSY[0] / 2.0, float(m_range[0]), float(m_range[1])], dtype=torch.float64, device=dev_)
                allt = [torch.zeros_like(mine_t) for _ in range(world)]
                dist.all_gather(allt, mine_t)
                ranges = [(int(t[1]), int(t[2]), float(t[0])) for t in allt]
                worst = max(t for _, _, t in ranges)
                rebalance_log.append(dict(ranges=[(a, b) for a, b, _ in ranges], step_ms=[1e3 * t for _, _, t in ranges]))
                if worst < best[0]:
                    best = (worst, [(a, b) for a, b, _ in ranges])
                if rnd == 3:
                    break
                m_range = parallel.rebalance_contiguous([(a, b) for a, b, _ in ranges], [t for _, _, t in ranges])
            m_range = best[1][rank]
        for _ in range(args.warmup):
            hot_path_step(tel, bt, kl, ctx, streams=args.streams, m_range=m_range, collect=collect)
        # A full (generation-2) cycle collection walks every object torch/numpy created at import
        # (~45 ms here) and would land at a random point of the timed region: collect now and move
        # the survivors to the permanent generation, as a long-running pipeline process would.
        import gc

        gc.collect()
        gc.freeze()
        for c in list(device._all):  # every context (main thread + stream workers)
            c.prof_reset(os.environ.get("DRIFT_BENCH_NOPROF") != "1")
        stage = []
        parallel.barrier()
        torch.cuda.synchronize()
        _BUSY[0] = 0.0
        t0 = time.perf_counter()
        for _ in range(args.steps):
            hot_path_step(tel, bt, kl, ctx, streams=args.streams, m_range=m_range, collect=collect)
        torch.cuda.synchronize()
        t_own = _BUSY[0]                      # this rank's own K steps: BT-gen + SVD + KL, without the waits in the collectives
        parallel.barrier()
        dt = time.perf_counter() - t0
        prof = {}
        for c in list(device._all):
            for k, v in c.prof_report().items():
                a = prof.setdefault(k, dict(ms=0.0, flops=0.0, launches=0))
                a["ms"] += v["ms"]; a["flops"] += v["flops"]; a["launches"] += v["launches"]
        # The per-stage wall times (`stage_ms`, `stages`) come from a few UNTIMED passes after the timed region: they need
        # a wait after BT-gen that the pipeline itself does not have (the host prepares the SVD stage while the GPU still
        # transforms), so their sum is a little above ms_per_step.
        # ... and they run at profiling level 2: every remaining kernel of the path is bracketed too (the "extended"
        # classes; their ~200 extra event pairs per step are kept out of the timed region)
        nstage = min(args.steps, 3)
        for c in list(device._all):
            c.prof_reset(2 if os.environ.get("DRIFT_BENCH_NOPROF") != "1" else 0)
        for _ in range(nstage):
            hot_path_step(tel, bt, kl, ctx, stage_times=stage, streams=args.streams, m_range=m_range, collect=collect)
        torch.cuda.synchronize()
        ext = {}
        for c in list(device._all):
            for k, v in c.prof_report().items():
                if k in EXT_CLASSES:
                    a = ext.setdefault(k, dict(ms=0.0, flops=0.0, launches=0))
                    a["ms"] += v["ms"]; a["flops"] += v["flops"]; a["launches"] += v["launches"]
        # BT-gen alone at both readings of the reference's SHT: plain quadrature (healpy iter = 0) and healpy's documented
        # default iter = 3 (the telescope's default, what the timed region ran)
        bt_ms = {}
        if rank == 0:
            from driftscan_amd import btgen as _btgen

            it_keep = tel.sht_iter
            for it in (0, 3):
                tel.sht_iter = it
                ts = []
                for _ in range(4):
                    ctx.sync(); torch.cuda.synchronize()
                    tb0 = time.perf_counter()
                    _bm = _btgen.beam_m_all(tel, ctx=ctx, m_range=m_range)
                    ctx.sync(); torch.cuda.synchronize()
                    ts.append(time.perf_counter() - tb0)
                    del _bm
                bt_ms[it] = 1e3 * min(ts[1:])
            tel.sht_iter = it_keep
        parallel.barrier()
        if world > 1 or force_dist:
            import torch.distributed as dist

            dev_ = "cuda" if args.backend == "nccl" else "cpu"
            tmine = torch.tensor([t_own, float(m_range[0] if m_range else 0), float(m_range[1] if m_range else nblocks - 1)],
                                 dtype=torch.float64, device=dev_)
            tall = [torch.zeros_like(tmine) for _ in range(world)]
            dist.all_gather(tall, tmine)
            rank_info = [dict(rank=r, step_ms=1e3 * float(t[0]) / args.steps, m_lo=int(t[1]), m_hi=int(t[2])) for r, t in enumerate(tall)]
            tmax = torch.tensor([dt], dtype=torch.float64, device=dev_)
            dist.all_reduce(tmax, op=dist.ReduceOp.MAX)
            dt = float(tmax.item())
        sharded = m_range is not None
        if not (world > 1 or force_dist):
            rank_info = [dict(rank=0, step_ms=1e3 * dt / args.steps, m_lo=m_range[0] if m_range else 0,
                              m_hi=m_range[1] if m_range else nblocks - 1)]
        value = (nblocks if sharded else world * nblocks) * args.steps / dt
        if args.shard:   # one rank of an emulated N-rank job: its own blocks over its own time
            value = (m_range[1] - m_range[0] + 1) * args.steps / dt
        if rank == 0:
            st = np.array(stage).mean(axis=0)
            trd_stride = int(ctx.lib.dm_prof_trd_stride())  # the column loop is sampled; the library scales the figures
            dom = max(prof, key=lambda k: prof[k]["ms"]) if prof else None
            roofline = None
            # HBM bytes per launch of the dominant kernel: PMC counters cannot be read from inside this
            # process, so the figure comes from the committed rocprofv3 --pmc passes of this same
            # command (profiles/*_pmc_traffic.json, made by scratch/run_profiles.sh + make_traffic_json.py).
            traffic, traffic_src, mfma_busy, valu_busy = None, None, None, None
            try:
                import glob

                bid = build_id()
                tj = sorted(glob.glob(os.path.join(ROOT, "profiles", "*_pmc_traffic.json")))
                if tj and dom is not None:
                    rec = json.load(open(tj[-1]))
                    if rec.get("_build_id") != bid:
                        # counters of another build of the kernels say nothing about this one
                        traffic_src = "%s is stale (build %s, running %s): traffic not reported" % (
                            os.path.relpath(tj[-1], ROOT), rec.get("_build_id"), bid)
                        rec = {}
                    key = {"zgemm_grouped": "zgemm_grouped_kernel<false, false>", "trd_symv": "trd_symv_kernel",
                           "trd_wx": "trd_wx_kernel", "gemm_grouped_realB": "zgemm_grouped_kernel<true, false>",
                           "dgemm_grouped": "dgemm_grouped_kernel", "jac_inner": "jac_inner_kernel<false>",
                           "jac_gram": "jac_gram_kernel", "jac_apply": "jac_apply_kernel",
                           "sb_chase": "sb_chase_pos_kernel", "sb_q2_apply": "sb_q2_apply_kernel<4>",
                           "sb_panel_qr": "sb_panel_fused_kernel"}.get(dom)
                    if dom == "zgemm_grouped" and key not in rec:
                        key = next((k for k in ("zgemm4_grouped_kernel<false, false, 1, 2>", "zgemm4_grouped_kernel<false, false, 1>")
                                    if k in rec), key)
                    if key is not None and key not in rec:  # kernels compiled inside a namespace (dm_trd32::trd_symv_kernel)
                        key = next((k for k in rec if k.endswith("::" + key)), key)
                    if key in rec:
                        traffic = rec[key]["fetch_bytes_per_launch"] + rec[key]["write_bytes_per_launch"]
                        traffic_src = os.path.relpath(tj[-1], ROOT)
                mj = sorted(f for f in glob.glob(os.path.join(ROOT, "profiles", "*_pmc_mfma.json")) if "configs2" not in os.path.basename(f))
                if mj:
                    mrec = json.load(open(mj[-1]))
                    if mrec.get("_build_id") == bid:
                        mfma_busy = dict(mrec, source=os.path.relpath(mj[-1], ROOT))
                    else:
                        mfma_busy = dict(stale="%s was taken at build %s, this is %s" % (os.path.relpath(mj[-1], ROOT),
                                                                                          mrec.get("_build_id"), bid))
                vj = sorted(glob.glob(os.path.join(ROOT, "profiles", "*_pmc_valu.json")))
                if vj:   # vector-ALU utilisation by counter: what the latency-bound classes (chase, jac_inner, q2, panel QR) have
                    vrec = json.load(open(vj[-1]))   # instead of a flop rate
                    if vrec.get("_build_id") == bid:
                        valu_busy = dict({k: (round(v["valu_busy"], 4) if isinstance(v, dict) else v) for k, v in vrec.items()},
                                         source=os.path.relpath(vj[-1], ROOT))
                    else:
                        valu_busy = dict(stale="%s was taken at build %s, this is %s" % (os.path.relpath(vj[-1], ROOT),
                                                                                          vrec.get("_build_id"), bid))
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
                                    sampling="every %dth launch of the column loop is bracketed by events (position "
                                             "walking through the panels); time, bytes and `launches` are the samples "
                                             "scaled by %d" % (trd_stride, trd_stride))
                else:
                    ach = p["flops"] / secs / 1e12 if secs > 0 else 0.0
                    roofline = dict(bound="mfma", kernel=dom, achieved=ach, peak=FP64_MFMA_PEAK_TFLOPS, unit="TFLOP/s",
                                    frac=ach / FP64_MFMA_PEAK_TFLOPS, traffic=traffic, traffic_unit="bytes/launch",
                                    traffic_source=traffic_src, launches=p["launches"],
                                    avg_launch_us=1e3 * p["ms"] / max(p["launches"], 1),
                                    flops_per_launch=p["flops"] / max(p["launches"], 1))
                    if dom in VALU_CLASSES:
                        roofline["pipe"] = ("fp64 VALU (the kernel issues no MFMA; on MI355X the fp64 vector peak equals the "
                                            "fp64 matrix peak, 78.6 TFLOP/s)")
                roofline["build_id"] = build_id()
                # every instrumented class, so that the cross-checks (sum of kernel time <= wall time) can be made; the
                # extended classes come from the untimed stage passes (profiling level 2)
                roofline["classes"] = class_table(prof, args.steps)
                roofline["classes"].update({k: dict(v, source="untimed stage passes (profiling level 2)")
                                            for k, v in class_table(ext, nstage).items()})
                roofline["classes_ms_sum"] = sum(v["ms_per_step"] for v in roofline["classes"].values())
                # the runner-up class of the other kind, for context (MFMA vs HBM side of the step)
                others = [k for k in prof if (k in HBM_CLASSES) != (dom in HBM_CLASSES)]
                if others:
                    o = max(others, key=lambda k: prof[k]["ms"])
                    q = prof[o]
                    if o in HBM_CLASSES:
                        oa = q["flops"] / (q["ms"] * 1e-3) / 1e9
                        roofline["also"] = dict(kernel=o, bound="hbm", achieved=oa, unit="GB/s", frac=oa / HBM_PEAK_GBS,
                                                ms_per_step=q["ms"] / args.steps)
                    else:
                        oa = q["flops"] / (q["ms"] * 1e-3) / 1e12
                        roofline["also"] = dict(kernel=o, bound="mfma", achieved=oa, unit="TFLOP/s",
                                                frac=oa / FP64_MFMA_PEAK_TFLOPS, ms_per_step=q["ms"] / args.steps)
                if mfma_busy is not None:
                    roofline["mfma_busy"] = mfma_busy
                    # the counter figure of the dominant class itself (the heaviest kernel of the class by MFMA cycles)
                    pref = {"zgemm_grouped": "zgemm4_grouped_kernel<false, false", "gemm_grouped_realB": "zgemm4_grouped_kernel<true, false",
                            "dgemm_grouped": "dgemm_grouped_kernel", "jac_gram": "jac_gram_kernel", "jac_apply": "jac_apply_kernel"}.get(dom)
                    cand = [v for k, v in mfma_busy.items() if pref and k.startswith(pref) and isinstance(v, dict) and "util" in v]
                    if cand:
                        roofline["mfma_busy_dominant"] = max(cand, key=lambda v: v.get("mfma_busy_cycles", 0.0))["util"]
                if valu_busy is not None:
                    roofline["valu_busy"] = valu_busy
            # per-stage fractions of SURVEY.md §8(d): algorithmic work of the stage / its wall time / fp64 MFMA peak
            my_ms = list(range(nblocks)) if m_range is None else list(range(m_range[0], m_range[1] + 1))
            WA, WB, WC = stage_work(tel, bt, my_ms, nkeep=dict(_NKEEP) if _NKEEP else None)
            stages = {}
            for name, W, secs in (("btgen", WA, st[0]), ("svd", WB, st[1]), ("kl", WC, st[2])):
                tf = W / secs / 1e12 if secs > 0 else 0.0
                stages[name] = dict(work_flop=W, ms=1e3 * secs, tflops=tf, frac_of_fp64_mfma_peak=tf / FP64_MFMA_PEAK_TFLOPS)
            cpu, parity = None, None
            if not args.no_cpu_baseline and world == 1:   # reported on rank 0 at N = 1 only
                keep, spec = {m: None for m in my_ms}, {}
                hot_path_step(tel, bt, kl, ctx, streams=args.streams, m_range=m_range, keep=keep, spectra=spec)   # untimed: real blocks
                cpu, parity = cpu_baseline(tel, bt, kl, {m: b for m, b in keep.items() if b is not None}, spec["sv"], spec["ev"])
                del keep
            line = {
                "metric": "m-blocks/sec (BT-gen + SVD + KL)",
                "value": value,
                "unit": "m-blocks/s",
                "n_gpus": world,
                "steps": args.steps,
                "warmup": args.warmup,
                "priming": {"passes": args.prime_passes, "seconds": t_prime, "note": "untimed passes of the same hot path "
                            "before the warm-up steps: the first ~1.5 s of GPU work on a fresh box run ~4 % slow"},
                "ms_per_step": 1e3 * dt / args.steps,
                "higher_is_better": True,
                "scaling": "strong" if sharded else "weak",
                "vs_baseline": None,
                "dtype": "f64",
                "data": "synthetic",
                "config": {"workload": "configs[1]: 32-feed unpolarised cylinder, nfreq=16, nbase=46, lmax=mmax=128, "
                                       + ("129 m-blocks split over the ranks in cost-balanced contiguous m-ranges"
                                          if sharded else "129 m-blocks per GPU per step") + ", KLTransform with foregrounds",
                           "nfreq": 16, "nbase": 46, "lmax": 128, "mmax": 128,
                           "sht_iter": int(tel.sht_iter),
                           "sht_note": "healpy.map2alm `iter` of the reference's SHT (through cora, not readable here): healpy's "
                                       "documented default 3, refined in harmonic space on the device; `btgen` carries both readings",
                           "sharding": "m-ranges, one job" if sharded else "m-blocks, one full workload per GPU",
                           "shard": None if not args.shard else "rank %s of an emulated sharded job alone on the GPU: m = %d..%d; "
                                    "`value` counts this rank's blocks only" % (args.shard, m_range[0], m_range[1]),
                           "mode": args.mode, "ranks": world, "backend": args.backend if world > 1 else None,
                           "collectives_in_timed_region": "gather of sigma/lambda spectra + all-reduce of a 9x9 band matrix"
                           if collect else None,
                           "streams_per_gpu": args.streams,
                           "kl_products": "all eigenvalues + every mode" if args.all_modes else
                           "all eigenvalues + the modes with S/N >= threshold (subset = True, what transform_save writes)"},
                "ranks": {"per_rank": rank_info, "rebalance": rebalance_log,
                          "imbalance_max_over_mean": max(r["step_ms"] for r in rank_info) / (sum(r["step_ms"] for r in rank_info) / len(rank_info)),
                          "note": "compute time of each rank per step (BT-gen + SVD + KL of its m-range, without the waits inside the "
                                  "collectives); `value` uses the MAX over ranks of the whole timed region.  Sharded mode: the step of a rank is a lock-step chain of a few hundred launches whose length "
                                  "follows the LARGEST matrix of its range, not the number of blocks — strong scaling of this small "
                                  "workload is bounded by that chain (DESIGN.md section 6)"},
                "stage_ms": {"btgen": 1e3 * st[0], "svd": 1e3 * st[1], "kl": 1e3 * st[2], "collectives": 1e3 * st[3]},
                "stages": stages,
                "btgen": {"btgen_iter0_ms": bt_ms.get(0), "btgen_iter3_ms": bt_ms.get(3),
                          "ratio": (bt_ms[3] / bt_ms[0]) if bt_ms.get(0) else None,
                          "note": "BT-gen of the step alone (all blocks of the rank, best of 3 after one warm-up call) with the "
                                  "plain equal-weight quadrature (iter 0) and with healpy's default three refinements (iter 3)"},
                # the two tridiagonalisation classes are timed on every 8th launch: scaled back to all launches
                "kernels_ms": {k: v["ms"] / args.steps for k, v in prof.items()},
                "roofline": roofline,
                "cpu_baseline": cpu,
                "parity": parity,
            }
        else:
            line = None
        ns_ok = not args.no_north_star and args.streams == 1 and not args.all_modes and not args.shard
        if world == 1 and not force_dist and ns_ok and rank == 0:
            del tel, bt, kl
            line["north_star"] = north_star_leg(args)
        elif world > 1 and ns_ok:
            del tel, bt, kl
            ns = north_star_job_leg(args, world, rank)   # every rank takes part: each starts its rank of the job
            if rank == 0:
                line["north_star"] = ns
        if rank == 0:
            # the full record goes to bench_detail.json and an EARLIER stdout line; the LAST line is the compact driver
            # line (< 8 KB: contract keys, roofline, cpu_baseline, parity, north_star in scalars)
            from benchlib import line as benchline

            benchline.emit(line)
    if world > 1 or force_dist:
        import torch.distributed as dist

        dist.barrier()
        dist.destroy_process_group()
    return 0


if __name__ == "__main__":
    sys.exit(main())
