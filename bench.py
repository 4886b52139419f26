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

Modes (m-blocks are independent: no data-path collective in either):
  weak     every rank runs the full 129-block workload (per-GPU work fixed); the singular-value and
           eigenvalue spectra of every rank are gathered to rank 0 inside the timed region, as
           `_collect_svd_spectrum` / `KLTransform._collect` do.  value = N * 129 * steps / time.
  sharded  ONE 129-block workload split over the ranks in contiguous m-ranges of equal estimated
           cost (`parallel.partition_contiguous`); each rank generates, compresses and transforms only
           its own blocks; the spectra gather and an all-reduce of a per-rank (nbands x nbands)
           matrix (the Fisher assembly pattern, psestimation.py:506-507) are inside the timed
           region.  value = 129 * steps / time ("scaling": "strong").

Prints ONE JSON line on rank 0.
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

CFG2 = dict(num_freq=16, freq_start=400.0, freq_end=450.0, freq_mode="edge", num_cylinders=2, cylinder_width=5.0,
            num_feeds=16, feed_spacing=0.4, tsys=1.0, force_lmax=128, force_mmax=128,
            sht_iter=3)   # healpy's documented default, stated explicitly: the measured configuration does not move with the library's default
FP64_MFMA_PEAK_TFLOPS = 78.6  # MI355X fp64 matrix peak (AMD datasheet)
HBM_PEAK_GBS = 8000.0         # HBM3E spec (MI355X_MICROARCH.md: 8 TB/s peak, ~6.3 achievable)
HBM_CLASSES = ("trd_symv", "trd_wx")
VALU_CLASSES = ("sb_panel_qr", "sb_chase", "sb_q2_apply")   # fp64 vector kernels of the two-stage tridiagonalisation
# every remaining kernel of the path, bracketed at profiling level 2 only (time, no work counter)
EXT_CLASSES = ("bt_ring", "bt_other", "trd_small", "dc", "chol_solve", "util", "eig_other", "svd_other")


def build_id():
    """Identity of the kernels this process runs: sha256 over the sources of libdriftmi (the GPU box has no .git).  The
    PMC records under profiles/ carry the id they were taken at; counters of another build are not reported."""
    import glob
    import hashlib

    h = hashlib.sha256()
    for f in sorted(glob.glob(os.path.join(ROOT, "driftscan_amd", "csrc", "*.h*")) + glob.glob(os.path.join(ROOT, "driftscan_amd", "csrc", "*.c"))):
        h.update(os.path.basename(f).encode())
        with open(f, "rb") as fh:
            h.update(fh.read())
    return h.hexdigest()[:16]


def parse_args(argv=None):
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=5)
    ap.add_argument("--warmup", type=int, default=2)
    ap.add_argument("--prime-passes", type=int, default=10,
                    help="untimed passes of the hot path before the warm-up steps (settles a fresh box; 0 to skip)")
    ap.add_argument("--mode", choices=["weak", "sharded"], default=None,
                    help="default: weak at --gpus 1 (one full workload), sharded at --gpus N > 1 (ONE workload in m-ranges)")
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
# CPU baseline (fresh process, no GPU): the WHOLE configs[1] job with the oracle on single-threaded workers
# ---------------------------------------------------------------------------------------------------
def host_cores():
    """Cores this process may actually use: the affinity mask, cut by the cgroup's CPU quota (a GPU box gives a
    one-GPU job a share of its host, not all 256 cores)."""
    try:
        n = len(os.sched_getaffinity(0))
    except Exception:
        n = os.cpu_count() or 1
    for path in ("/sys/fs/cgroup/cpu.max", "/sys/fs/cgroup/cpu/cpu.cfs_quota_us"):
        try:
            txt = open(path).read().split()
            if path.endswith("cpu.max"):
                if txt[0] != "max":
                    n = min(n, max(1, int(round(float(txt[0]) / float(txt[1])))))
            else:
                q = float(txt[0])
                per = float(open("/sys/fs/cgroup/cpu/cpu.cfs_period_us").read())
                if q > 0:
                    n = min(n, max(1, int(round(q / per))))
            break
        except Exception:
            continue
    return max(1, n)


def _cpu_one_block(job):
    """SVD chain + covariance projections + KL of one m-block with the oracle: per-stage seconds, the spectra, and the
    noise covariance (kept by the caller for the conditioning bound of the pencil, computed outside the timing)."""
    import numpy as np

    from oracle import kl as okl
    from oracle import svdchain as osvd

    blk, noisew, cv_sg, cv_fg, npw, polsvcut, svcut = job
    t0 = time.perf_counter()
    o = osvd.svd_m(blk, noisew, polsvcut=polsvcut)
    t1 = time.perf_counter()
    if int(osvd.svd_num(o["singularvalues"], svcut)[0].sum()) == 0:
        # no mode above svcut (an m beyond the telescope's band limit): the reference's `nside == 0` early-out, kltransform.py:324-326
        return t1 - t0, 0.0, 0, np.asarray(o["singularvalues"]), np.zeros(0), np.zeros((0, 0), dtype=np.complex128)
    cs, cn = okl.sn_covariance(o["beam_svd"], o["beam_ut"], o["singularvalues"], cv_sg, cv_fg, npw, svcut=svcut)
    ev = okl.kl_transform_m(cs, cn)[0]
    t2 = time.perf_counter()
    return t1 - t0, t2 - t1, int(cs.shape[0]), np.asarray(o["singularvalues"]), np.asarray(ev), cn


def _cpu_bt_columns(desc):
    from oracle import btgen as ob

    t0 = time.perf_counter()
    ob.beam_transfer_m(desc)
    return time.perf_counter() - t0


def _cpu_worker(wid, tasks, results, shared):
    """One single-threaded rank of the CPU job (the reference's MPI mode, `OMP_NUM_THREADS=1`): takes tasks from the
    common queue until it is empty, reports when it ran dry, THEN (untimed) the conditioning bounds of the pencils it solved."""
    import numpy as np

    jobs, common, desc0 = shared
    stash = {}
    try:
        while True:
            t = tasks.get()
            if t is None:
                break
            if t[0] == "block":
                m = t[1]
                ts, tk, ndof, sv, ev, cn = _cpu_one_block((jobs[m],) + common)
                stash[m] = cn
                results.put(("block", m, ts, tk, ndof, sv, ev))
            else:   # ("bt", fi, b0, b1): the (f, b) columns of one frequency and a range of baselines, all m
                desc = dict(desc0, included_freq=np.array([t[1]]), included_baseline=np.arange(t[2], t[3]))
                results.put(("bt", t[1], t[2], t[3], _cpu_bt_columns(desc)))
        results.put(("dry", wid, time.perf_counter()))
        for m, cn in stash.items():
            tol = 1e-10
            if cn.shape[0]:
                w = np.linalg.eigvalsh(0.5 * (cn + cn.conj().T))
                tol = max(1e-10, 50.0 * 2.220446049250313e-16 * abs(w[-1]) / max(abs(w[0]), 1e-300))   # tests/parity_util.pencil_tol
            results.put(("tol", m, tol))
    except BaseException as e:   # the parent must hear about it: it counts "end" messages
        import traceback

        results.put(("error", wid, "%r\n%s" % (e, traceback.format_exc()[-1500:])))
    finally:
        results.put(("end", wid))


def cpu_worker_main(path, nproc):
    """`bench.py --cpu-worker file nproc`: the whole job on `nproc` single-threaded worker processes over a common task
    queue (m-blocks largest first, BT-gen column chunks in between); wall = start to the moment the last worker ran dry."""
    os.environ["OMP_NUM_THREADS"] = os.environ["OPENBLAS_NUM_THREADS"] = os.environ["MKL_NUM_THREADS"] = "1"
    import multiprocessing as mp
    import pickle

    import numpy as np

    with open(path, "rb") as fh:
        jobs, common, desc0, bt_tasks = pickle.load(fh)
    nproc = int(nproc)
    ctxm = mp.get_context("fork")
    tasks, results = ctxm.Queue(), ctxm.Queue()
    order = sorted(jobs)                       # ndof (and the cost) falls with m: largest first
    tl = [("block", m) for m in order]
    step = max(1, len(tl) // max(len(bt_tasks), 1))
    merged, bi = [], 0
    for i, t in enumerate(tl):                 # BT-gen chunks spread through the first part of the queue
        if bi < len(bt_tasks) and i % step == 0:
            merged.append(("bt",) + tuple(bt_tasks[bi])); bi += 1
        merged.append(t)
    merged += [("bt",) + tuple(t) for t in bt_tasks[bi:]]
    for t in merged:
        tasks.put(t)
    for _ in range(nproc):
        tasks.put(None)
    t0 = time.perf_counter()
    procs = [ctxm.Process(target=_cpu_worker, args=(w, tasks, results, (jobs, common, desc0))) for w in range(nproc)]
    for p_ in procs:
        p_.start()
    blocks, bts, tols, dry, ended, errors = {}, [], {}, [], 0, []
    import queue as _queue

    while ended < nproc:
        try:
            r = results.get(timeout=30.0)
        except _queue.Empty:
            if not any(p_.is_alive() for p_ in procs):   # everybody gone without saying so (killed): do not wait for ever
                errors.append("worker processes died")
                break
            continue
        if r[0] == "error":
            errors.append(r[2])
        elif r[0] == "block":
            blocks[r[1]] = r[2:]
        elif r[0] == "bt":
            bts.append(r[4])
        elif r[0] == "dry":
            dry.append(r[2] - t0)
        elif r[0] == "tol":
            tols[r[1]] = r[2]
        else:
            ended += 1
    for p_ in procs:
        p_.join(10.0)
        if p_.is_alive():
            p_.kill()   # exactly the processes started above
    if errors or len(blocks) != len(jobs):
        print("cpu worker failed: %s" % (errors[:1] or ["%d of %d blocks done" % (len(blocks), len(jobs))]), file=sys.stderr)
        sys.exit(3)
    out = dict(wall_s=max(dry), workers=nproc, dry_s=dry, bt_core_s=float(sum(bts)), bt_tasks=len(bts),
               svd_core_s=float(sum(v[0] for v in blocks.values())), kl_core_s=float(sum(v[1] for v in blocks.values())))
    with open(path + ".out", "wb") as fh:
        pickle.dump((out, {m: (v[2], v[3], v[4]) for m, v in blocks.items()}, tols), fh)
    print(json.dumps(out))


def cpu_baseline(tel, bt, kl, blocks, gpu_sv, gpu_ev):
    """The oracle (numpy/scipy restatement, kind = "port") on the host cores: the WHOLE configs[1] job, MEASURED — the
    SVD chain + covariance projections + KL of ALL 129 real m-blocks (`blocks`: {m: (F,2,B,P,L) numpy}, copied back from
    the device) and the BT-gen of every (f, b) column (pixel kernels, one FFT per ring, Legendre matrix products), on
    single-threaded worker processes over m (the reference's MPI mode, `OMP_NUM_THREADS=1` per rank) — as many as this
    process may use cores (`host_cores`).  value = blocks / wall; nothing is extrapolated.  The spectra the oracle
    computes are compared with the GPU's (`gpu_sv[m]` (F, K), `gpu_ev[m]` (ndof,)): the `parity` object of the line."""
    import pickle
    import tempfile

    import numpy as np
    import scipy

    ncores = host_cores()
    M = tel.mmax + 1
    ms = sorted(blocks)
    nproc = max(1, min(ncores, len(ms)))
    desc0 = dict(polarised=False, zenith=tel.zenith, baselines=tel.baselines, uniquepairs=tel.uniquepairs,
                 beamclass=tel.beamclass, wavelengths=tel.wavelengths, cylinder_width=tel.cylinder_width,
                 fwhm_e=tel.fwhm_e, fwhm_h=tel.fwhm_h, lmax=tel.lmax, mmax=tel.mmax, l_boost=tel.l_boost,
                 included_freq=np.array([0]), included_baseline=np.array([0]),
                 accuracy_boost=tel.accuracy_boost, sht_iter=tel.sht_iter, sht_fft=True)   # one FFT per ring, Legendre sums as matrix products
    half = (tel.nbase + 1) // 2
    bt_tasks = [(fi, b0, min(b0 + half, tel.nbase)) for fi in range(tel.nfreq) for b0 in range(0, tel.nbase, half)]
    noisew = bt._noisew()[:, : tel.nbase]
    common = (noisew, kl.signal(), kl.foreground(), kl._npower(1.0), bt.polsvcut, bt.svcut)
    # fresh process (this one holds a GPU context: never fork or exec from it)
    try:
        with tempfile.TemporaryDirectory() as tmp:
            path = os.path.join(tmp, "jobs.pkl")
            with open(path, "wb") as fh:
                pickle.dump((blocks, common, desc0, bt_tasks), fh, protocol=4)
            env = dict(os.environ, OMP_NUM_THREADS="1", OPENBLAS_NUM_THREADS="1", MKL_NUM_THREADS="1")
            subprocess.run([sys.executable, os.path.abspath(__file__), "--cpu-worker", path, str(nproc)],
                           env=env, stdout=subprocess.PIPE, timeout=1200, check=True)
            with open(path + ".out", "rb") as fh:
                w, spectra, tols = pickle.load(fh)
    except Exception as e:  # the baseline is reporting only: never lose the bench line over it
        return dict(value=0.0, unit="m-blocks/s", cores=ncores, kind="port", error=repr(e), sample="failed"), None
    # ---- parity of the GPU spectra against the oracle's, every block
    sv_err, ev_err, ev_over, svnum_eq, kept_eq, kept_escape, worst = 0.0, 0.0, 0.0, True, 0, 0, None
    kept_rel, kept_n, kept_worst = 0.0, 0, None   # element-wise relative error of the KEPT eigenvalues (lambda_o >= threshold)
    for m in ms:
        ndof, sv_o, ev_o = spectra[m]
        sv_g, ev_g = np.asarray(gpu_sv[m]), np.asarray(gpu_ev[m])
        if sv_o.max() > 0:
            sv_err = max(sv_err, float(np.abs(sv_g - sv_o).max() / sv_o.max()))
        n_g = (sv_g > sv_g.max() * bt.svcut).sum(axis=1) if sv_g.max() > 0 else np.zeros(sv_g.shape[0], int)
        n_o = (sv_o > sv_o.max() * bt.svcut).sum(axis=1) if sv_o.max() > 0 else np.zeros(sv_o.shape[0], int)
        svnum_eq = svnum_eq and bool(np.array_equal(n_g, n_o))
        if ev_o.size and ev_g.shape == ev_o.shape:
            lam = float(np.abs(ev_o).max())
            e = float(np.abs(ev_g - ev_o).max() / max(lam, 1e-300))
            tol = tols.get(m, 1e-10)
            if e / tol > ev_over:
                ev_over, worst = e / tol, dict(m=int(m), ndof=int(ndof), err=e, pencil_tol=tol)
            ev_err = max(ev_err, e)
            kp = ev_o >= kl.threshold          # the modes transform_save keeps (kltransform.py:385-398): what goes downstream
            if kp.any():
                r_ = np.abs(ev_g[kp] - ev_o[kp]) / ev_o[kp]
                kept_n += int(kp.sum())
                if float(r_.max()) > kept_rel:
                    kept_rel, kept_worst = float(r_.max()), dict(m=int(m), ndof=int(ndof), lambda_o=float(ev_o[kp][int(r_.argmax())]))
            kg, ko = int((ev_g >= kl.threshold).sum()), int((ev_o >= kl.threshold).sum())
            if kg == ko:
                kept_eq += 1
            elif np.abs(ev_o - kl.threshold).min() <= tol * lam:
                kept_escape += 1
        elif ev_g.shape != ev_o.shape:
            svnum_eq = False
        else:
            kept_eq += 1
    # Blocks whose eigenvalues differ from the oracle's by more than pencil_tol: the conditioning bound of the PENCIL does
    # not cover the svcut truncation in front of it (a kept subspace with a singular value close to the cut is only
    # determined to eps sigma_1 / gap).  Their yardstick is the sensitivity of the oracle's OWN answer: its whole chain run
    # again on the block perturbed by one unit roundoff per entry (as tests/parity_util.pencil_sensitivity does for a pencil).
    over = []
    for m in ms:
        ndof, sv_o, ev_o = spectra[m]
        ev_g = np.asarray(gpu_ev[m])
        if ev_o.size and ev_g.shape == ev_o.shape:
            e = float(np.abs(ev_g - ev_o).max() / max(float(np.abs(ev_o).max()), 1e-300))
            if e > tols.get(m, 1e-10):
                over.append((e / tols.get(m, 1e-10), m, e))
    over.sort(reverse=True)
    sens_rec, over_ok = [], True
    rng = np.random.default_rng(12345)
    for _, m, e in over[:8]:
        ev_o = spectra[m][2]
        worst_s = 0.0
        for _rep in range(2):
            blk = blocks[m]
            pert = blk * (1.0 + 2.220446049250313e-16 * rng.standard_normal(blk.shape)) \
                + 1j * blk.imag * (2.220446049250313e-16 * rng.standard_normal(blk.shape))
            ev_p = _cpu_one_block((pert,) + common)[4]
            worst_s = float("inf") if ev_p.shape != ev_o.shape else max(
                worst_s, float(np.abs(ev_p - ev_o).max() / max(float(np.abs(ev_o).max()), 1e-300)))
        ok = e <= max(tols.get(m, 1e-10), 10.0 * worst_s)
        over_ok = over_ok and ok
        sens_rec.append(dict(m=int(m), ndof=int(spectra[m][0]), err=e, pencil_tol=tols.get(m, 1e-10),
                             oracle_sensitivity_to_one_ulp_of_the_block=worst_s, within_10x_sensitivity=bool(ok)))
    if len(over) > 8:
        over_ok = False   # more offenders than were examined: not claimed
    parity = dict(blocks=len(ms), sv_max_err_over_svmax=sv_err, sv_tol=1e-10, svnum_equal=svnum_eq,
                  ev_blocks_over_pencil_tol=len(over), ev_over_pencil_tol_examined=sens_rec,
                  ev_max_err_over_lambda_max=ev_err, ev_max_err_over_pencil_tol=ev_over, ev_worst=worst,
                  ev_kept_max_rel_err=kept_rel, ev_kept_modes=kept_n, ev_kept_worst=kept_worst, ev_kept_rel_tol=1e-4,
                  ev_kept_note="max over the modes with lambda_o >= threshold of |lambda - lambda_o| / lambda_o, element-wise: the "
                               "reference's own bar is rel 1e-4 (tests/test_functional.py:29-31,209); north_star asks 1e-10",
                  kept_counts_equal=kept_eq, kept_counts_differ_with_an_eigenvalue_within_tol_of_the_cut=kept_escape,
                  kept_counts_differ_otherwise=len(ms) - kept_eq - kept_escape,
                  green=bool(sv_err <= 1e-10 and svnum_eq and (ev_over <= 1.0 or over_ok) and kept_eq + kept_escape == len(ms)
                             and kept_rel <= 1e-4),
                  green_rule="sigma within 1e-10 sigma_max, svnum equal, kept counts equal (or an eigenvalue within tol of the cut), kept "
                             "eigenvalues element-wise within rel 1e-4, "
                             "eigenvalues within pencil_tol — or, for the blocks beyond it, within 10 x the measured sensitivity of "
                             "the oracle's own spectrum to a one-ulp perturbation of the block",
                  note="GPU spectra of the timed configuration against the oracle's on the SAME real blocks, all %d of them: "
                       "singular values relative to the block's largest (bound 1e-10), eigenvalues relative to lambda_max "
                       "against pencil_tol = max(1e-10, 50 eps cond(N)) (tests/parity_util.py), svnum and kept-mode counts" % len(ms))
    core_s = w["bt_core_s"] + w["svd_core_s"] + w["kl_core_s"]
    return dict(value=M / w["wall_s"], unit="m-blocks/s", extrapolated=False, cores=nproc, kind="port", mode="workers",
                wall_s=w["wall_s"], host_cores=ncores, core_seconds=dict(btgen=w["bt_core_s"], svd=w["svd_core_s"], kl=w["kl_core_s"],
                                                                         total=core_s),
                parallel_efficiency=core_s / (w["wall_s"] * nproc),
                stage_s_per_block_one_core=dict(btgen=w["bt_core_s"] / M, svd=w["svd_core_s"] / M, kl=w["kl_core_s"] / M),
                sample="the WHOLE job, measured: oracle (numpy %s / scipy %s) SVD chain + covariance projections + KL of all %d "
                       "real configs[1] blocks copied back from the device and BT-gen (pixel kernels, one FFT per ring, Legendre "
                       "matrix products) of all %d (f, b) columns, on %d single-threaded worker processes over one task queue "
                       "(the reference's MPI mode); wall = start to the last worker running dry"
                       % (np.__version__, scipy.__version__, len(ms), tel.nfreq * tel.nbase, nproc)), parity


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


def stage_work(tel, bt, ms, nkeep=None):
    """Algorithmic work per stage, SURVEY.md §8(d): W_A (Legendre, 8 Nr Lm F B P per m), W_B (SVD chain),
    W_C (covariance projections + eig), summed over the given m."""
    import numpy as np

    from driftscan_amd import healpix

    F, B, P, L = tel.nfreq, tel.nbase, tel.num_pol_sky, tel.lmax + 1
    T = 2 * B
    lmax_bf, _ = tel.baseline_lmax(np.arange(B), np.full(B, F - 1))
    nside = healpix.nside_for_lmax(int(lmax_bf.max()), tel.accuracy_boost if P == 1 else 1)
    Nr = 4 * nside - 1

    def svd(a, b):
        lo, hi = min(a, b), max(a, b)
        return 4.0 * (2.0 * hi * lo * lo + 11.0 * lo ** 3)

    WA = WB = WC = 0.0
    for mi in ms:
        Lm = L - mi
        WA += 8.0 * Nr * Lm * F * B * P
        svnum = bt._svd_num(mi)[0]
        ndof = float(svnum.sum())
        for f in range(F):
            n = float(svnum[f])
            if P == 1:
                WB += svd(T, Lm) + svd(n, Lm) + 8.0 * T * Lm * n
            else:
                r1 = r2 = float(min(T, P * Lm))
                WB += svd(T, P * Lm) + svd(r1, (P - 1) * Lm) + svd(r2, Lm) + svd(n, P * Lm) \
                    + 8.0 * T * P * Lm * (r1 + r2 + n) + 8.0 * T * (r2 * r1 + n * r2)
        nF = 1 if P == 1 else 3
        # eig(n) of SURVEY.md section 8(d) = 4 (n^3/3 potrf + n^3 hegst + 4n^3/3 hetrd + 2n^3 back-transform + n^3
        # back-solve) = 68 n^3 / 3 forms EVERY eigenvector; only the nkeep modes that are kept are back-transformed
        # here, so the EXECUTED work is counted: the last two terms scale with nkeep / n
        nk = float(nkeep.get(mi, ndof)) if nkeep is not None else ndof
        eig = 4.0 * (ndof ** 3 / 3.0 + ndof ** 3 + 4.0 * ndof ** 3 / 3.0 + 3.0 * ndof * ndof * nk)
        WC += 8.0 * ndof * ndof * Lm * (1 + nF) + 8.0 * T * float((svnum.astype(np.float64) ** 2).sum()) + eig
    return WA, WB, WC


CFG3 = dict(num_freq=64, freq_start=400.0, freq_end=500.0, freq_mode="edge", num_cylinders=4, cylinder_width=12.0,
            num_feeds=16, feed_spacing=0.4, tsys=1.0, force_lmax=512, force_mmax=512, sht_iter=3)


def class_table(pr, steps=1.0):
    """Per-class figures of a dm_prof_report: ms, launches, algorithmic rate and the fraction of the peak that bounds the
    class (fp64 MFMA / VALU 78.6 TFLOP/s, HBM 8 TB/s); the extended classes carry time only."""
    out = {}
    for k, v in pr.items():
        hbm = k in HBM_CLASSES
        rate = (v["flops"] / (v["ms"] * 1e-3) / (1e9 if hbm else 1e12)) if (v["ms"] > 0 and v["flops"] > 0) else None
        out[k] = dict(ms_per_step=v["ms"] / steps, launches_per_step=v["launches"] / steps, rate=rate,
                      unit="GB/s" if hbm else "TFLOP/s",
                      frac=None if rate is None else rate / (HBM_PEAK_GBS if hbm else FP64_MFMA_PEAK_TFLOPS))
    return out


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


CFG5 = dict(num_freq=256, freq_start=400.0, freq_end=800.0, freq_mode="edge", num_cylinders=4, cylinder_width=14.5,
            num_feeds=64, feed_spacing=0.3, tsys=1.0, force_lmax=1024, force_mmax=1024, sht_iter=3)


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


def _share_child(args, share):
    """One emulated share of the configs[2] job in a child process (started, never exec'ed into: this process keeps its HIP
    context; a failure of the leg — a share holds ~150 GB of HBM — must not cost the configs[1] line)."""
    retried = None
    cmd = [sys.executable, os.path.abspath(__file__), "--workload", "configs2", "--share", share]
    res = subprocess.run(cmd, stdout=subprocess.PIPE, stderr=subprocess.PIPE, timeout=600)
    if res.returncode != 0 and "DRIFT_BENCH_SVD_GB" not in os.environ:
        # once more at the batch budgets of rounds 1-3 (125 / 48 / 48 / 80 GB): a card with less free memory than the
        # 230 GB the default budgets take should still give a figure — the line says which budgets it ran with
        retried = res.stderr.decode()[-300:]
        env = dict(os.environ, DRIFT_BENCH_BEAM_GB="125", DRIFT_BENCH_SVD_GB="48", DRIFT_BENCH_KL_GB="48", DRIFTMI_WORKSPACE_GB="80")
        res = subprocess.run(cmd, stdout=subprocess.PIPE, stderr=subprocess.PIPE, timeout=600, env=env)
    if res.returncode != 0:
        raise RuntimeError("share %s exited with %d: %s" % (share, res.returncode, res.stderr.decode()[-400:]))
    return json.loads(res.stdout.decode().strip().splitlines()[-1]), retried


def committed_shares():
    """The latest profiles/*_configs2_shares.json: ALL N shares of the configs[2] job measured on one GPU each in one
    gpurun call (scratch/shares_all.sh); None when absent."""
    import glob

    fl = sorted(glob.glob(os.path.join(ROOT, "profiles", "*_configs2_shares.json")))
    if not fl:
        return None
    try:
        rec = json.load(open(fl[-1]))
        rec["_file"] = os.path.relpath(fl[-1], ROOT)
        return rec
    except Exception:
        return None


def north_star_leg(args):
    """The north-star workload inside the default line (rank 0 at --gpus 1): shares of the BASELINE configs[2] job through
    ProductManager.generate(), every kernel class timed, stage-level W / t.  Measured LIVE: share 0/8 (lowest m: largest
    matrices) and the share the committed all-shares record names as the slowest; `projected_job_s` = the MAX over the
    eight shares — live figures where this run has them, the committed record (made at the build it names) for the rest."""
    import gc

    import torch

    from driftscan_amd import beamtransfer, device

    beamtransfer.BeamTransfer._clcache.clear()
    device.reset_context()
    gc.collect()
    torch.cuda.empty_cache()
    t0 = time.perf_counter()
    rec = committed_shares()
    n = int(args.north_star_share.split("/")[1])
    shares = [args.north_star_share]
    if rec and rec.get("n") == n and rec.get("shares") and os.environ.get("DRIFT_BENCH_NS_ONE") != "1":
        slow = max(rec["shares"], key=lambda r_: r_["share_s"])["share"]
        if slow not in shares:
            shares.append(slow)
    live, retried = {}, None
    try:
        for k_, sh_ in enumerate(shares):
            # the driver wipes what the previous process freed in the background, and the first large allocations of the
            # next one wait for it (scratch/shares_all.py: 2-3 s in front of the BT-gen kernels, once an out-of-memory);
            # a rank of a real job starts on an idle card
            time.sleep(float(os.environ.get("DRIFT_BENCH_NS_PAUSE", "10")))
            live[sh_], rt = _share_child(args, sh_)
            retried = retried or rt
    except Exception as e:   # reporting only
        if not live:
            return dict(error=repr(e))
    sh = live[shares[0]]
    keep = ("share_s", "share_note", "classes", "kernel_s", "kernel_coverage_of_wall", "stages", "zgemm_cov", "hbm_peak_gb",
            "m_range")
    out = {k: sh.get(k) for k in keep}
    out["workload"] = sh["config"]["workload"]
    out["budgets_gb"] = sh["config"].get("budgets_gb")
    if retried is not None:
        out["first_attempt_failed"] = retried
    out["share"] = shares[0]
    out["sht_iter"] = sh["config"]["sht_iter"]
    out["m_blocks"] = sh["value"] * sh["share_s"]
    # every share of the job: live where measured now, else the committed record
    bid = build_id()
    allsh = {}
    if rec and rec.get("n") == n:
        for r_ in rec["shares"]:
            allsh[r_["share"]] = dict(share_s=r_["share_s"], m_range=r_.get("m_range"), source=rec["_file"],
                                      build_id=rec.get("_build_id"), stale=rec.get("_build_id") != bid)
    for k, v in live.items():
        allsh[k] = dict(share_s=v["share_s"], m_range=v.get("m_range"), source="live", build_id=bid, stale=False,
                        stages={kk: vv["seconds"] for kk, vv in (v.get("stages") or {}).items() if isinstance(vv, dict)})
    out["shares"] = allsh
    complete = len(allsh) == n
    worst = max(allsh, key=lambda k: allsh[k]["share_s"])
    out["projected_job_s"] = allsh[worst]["share_s"] if complete else None
    out["projected_job_slowest_share"] = worst if complete else None
    out["job_m_blocks_per_s"] = (sh["config"]["mmax"] + 1) / allsh[worst]["share_s"] if complete else None
    out["projected_job_note"] = ("MAX over the %d shares of the cost-balanced partition (m-blocks are independent, no data-path "
                                 "collective); %d measured in this run, the others from %s%s" % (
                                     n, len(live), rec["_file"] if rec else "nothing (no committed all-shares record)",
                                     " — STALE build for those" if any(v["stale"] for v in allsh.values()) else ""))
    if len(shares) > 1 and shares[1] in live:
        s2 = live[shares[1]]
        out["second_share"] = dict(share=shares[1], share_s=s2["share_s"], m_range=s2.get("m_range"), kernel_s=s2["kernel_s"],
                                   stages=s2.get("stages"), zgemm_cov=s2.get("zgemm_cov"))
    # counter evidence at THIS workload (rocprofv3 --pmc restricted to the kernels of interest, scratch/pmc_share.sh)
    try:
        import glob

        pj = sorted(glob.glob(os.path.join(ROOT, "profiles", "*_configs2_pmc_mfma.json")))
        if pj and out.get("zgemm_cov"):
            pr = json.load(open(pj[-1]))
            if pr.get("_build_id") == bid:
                out["zgemm_cov"]["mfma_busy"] = pr.get("zgemm_cov", {}).get("mfma_busy")
                out["zgemm_cov"]["mfma_busy_source"] = os.path.relpath(pj[-1], ROOT)
                out["pmc"] = {k: v for k, v in pr.items() if not k.startswith("_")}
            else:
                out["zgemm_cov"]["mfma_busy"] = None
                out["zgemm_cov"]["mfma_busy_source"] = "%s is stale (build %s, running %s)" % (
                    os.path.relpath(pj[-1], ROOT), pr.get("_build_id"), bid)
    except Exception:
        pass
    import glob as _glob

    cjs = sorted(_glob.glob(os.path.join(ROOT, "profiles", "r*_configs2_cpu_sample.json")))
    if cjs:   # the oracle on real configs[2] blocks (scratch/cpu_sample_configs2.py): seconds per sample AND its sigma against the device's
        try:
            out["cpu_sample"] = dict(json.load(open(cjs[-1])), source=os.path.relpath(cjs[-1], ROOT), live=False)
        except Exception:
            pass
    out["leg_wall_s"] = time.perf_counter() - t0
    out["target"] = "full configs[2] product set in under 600 s on 8 x MI355X; covariance GEMMs at >= 0.5 of the fp64 MFMA peak"
    return out


def north_star_job_leg(args, world, rank):
    """The north-star workload at --gpus N > 1: the REAL N-rank configs[2] job (`measure_job`).  Every rank process of the
    configs[1] line starts ONE child — its rank of the job, on its GPU, in a fresh process group on a port rank 0 picks —
    and waits for it; rank 0's child prints the job's line."""
    import gc
    import socket

    import torch

    from driftscan_amd import beamtransfer, device, parallel

    beamtransfer.BeamTransfer._clcache.clear()
    device.reset_context()
    gc.collect()
    torch.cuda.empty_cache()
    port = None
    if rank == 0:
        sk = socket.socket()
        sk.bind(("127.0.0.1", 0))
        port = sk.getsockname()[1]
        sk.close()
    port = parallel.bcast_object(port)
    # a launcher's elastic agent variables would send the child's rendezvous to the PARENT job's store
    env = {k: v for k, v in os.environ.items() if not k.startswith("TORCHELASTIC_")}
    env.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world),
               LOCAL_RANK=os.environ.get("LOCAL_RANK", str(rank)), HSA_ENABLE_IPC_MODE_LEGACY="0")
    cmd = [sys.executable, os.path.abspath(__file__), "--workload", "configs2", "--job", "--gpus", str(world),
           "--backend", args.backend] + (["--one-gpu"] if args.one_gpu else []) + (
               ["--share-mmax", str(args.share_mmax)] if args.share_mmax else [])
    t0 = time.perf_counter()
    out = None
    try:
        res = subprocess.run(cmd, stdout=subprocess.PIPE, stderr=subprocess.PIPE, timeout=900, env=env)
        if rank == 0:
            if res.returncode != 0:
                out = dict(error="rank 0 of the job exited with %d: %s" % (res.returncode, res.stderr.decode()[-600:]))
            else:
                out = dict(job=json.loads(res.stdout.decode().strip().splitlines()[-1]))
    except Exception as e:
        if rank == 0:
            out = dict(error=repr(e))
    parallel.barrier()
    if rank == 0:
        out["leg_wall_s"] = time.perf_counter() - t0
        out["what"] = ("the REAL %d-rank BASELINE configs[2] job%s through ProductManager.generate(), one child process per rank "
                       "started by the rank processes of this line" % (world, " (toy-telescope REHEARSAL)" if args.share_mmax else ""))
        out["target"] = "full configs[2] product set in under 600 s on 8 x MI355X; covariance GEMMs at >= 0.5 of the fp64 MFMA peak"
    return out


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
        args.mode = "sharded" if args.gpus > 1 else "weak"
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
                           "sb_chase": "sb_chase2_kernel<4>", "sb_q2_apply": "sb_q2_apply_kernel<4>",
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
