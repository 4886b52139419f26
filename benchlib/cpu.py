"""The CPU baseline leg of bench.py: the oracle (numpy / scipy restatement of the reference's per-m path, kind = "port") on the
host cores — the WHOLE configs[1] job, measured — and the parity of the GPU spectra against it.  The ONLY place outside tests/
and __graft_entry__.smoke() that touches `oracle/`, and only as the thing compared with and timed beside, never as the product."""
import json
import os
import subprocess
import sys
import time

from .common import BENCH, host_cores

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
            subprocess.run([sys.executable, BENCH, "--cpu-worker", path, str(nproc)],
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
