#!/usr/bin/env python3
"""A SAMPLED reference-side CPU figure at the north-star size (BASELINE configs[2]: 128-feed polarised cylinder, nfreq 64,
lmax 512): the oracle (numpy/scipy restatement of the reference's per-m path) timed on the GPU box's host on REAL blocks
made by the device — the SVD chain of 2 frequencies of m = 0 and of m = 460 (one core each), the covariance projections +
KL of m = 460 — and scaled to the whole job with the product's own cost model.  Everything about the scaling is an
ESTIMATE and is labelled so; the measured numbers are the per-sample seconds.

    python scratch/cpu_sample_configs2.py --out gpurun_out/r05_configs2_cpu_sample.json
"""
import argparse
import json
import os
import sys
import tempfile
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--out", required=True)
    args = ap.parse_args()
    os.environ["OMP_NUM_THREADS"] = os.environ["OPENBLAS_NUM_THREADS"] = os.environ["MKL_NUM_THREADS"] = "1"
    import scipy

    from driftscan_amd import beamtransfer, btgen, cylinder, device, kltransform
    from oracle import kl as okl
    from oracle import svdchain as osvd

    ctx = device.get_context(workspace_bytes=40 << 30)
    tel = cylinder.PolarisedCylinderTelescope.from_config(dict(bench.CFG3))
    L = tel.lmax + 1
    out = dict(workload="BASELINE configs[2] (nfreq 64, nbase %d, lmax = mmax 512): oracle on one host core, sampled" % tel.nbase,
               numpy=np.__version__, scipy=scipy.__version__, host_cores=bench.host_cores(), threads_per_sample=1, samples={})
    with tempfile.TemporaryDirectory() as tmp:
        bt = beamtransfer.BeamTransfer(tmp, telescope=tel)
        kl = kltransform.KLTransform.from_config(dict(threshold=0.1), bt, subdir="kl")
        noisew = bt._noisew()[:, : tel.nbase]
        for m in (0, 460):
            blk_d = btgen.beam_m_all(tel, ctx=ctx, max_bytes=24 << 30, m_range=(m, m))
            ctx.sync()
            fsel = [0, tel.nfreq // 2]
            blk2 = blk_d[0][fsel].cpu().numpy()                       # (2, 2, B, P, L)
            t0 = time.perf_counter()
            o2 = osvd.svd_m(blk2, noisew[fsel], polsvcut=bt.polsvcut)
            t_svd2 = time.perf_counter() - t0
            rec = dict(svd_chain_s_for_2_frequencies=t_svd2, svd_chain_s_per_frequency=t_svd2 / 2.0,
                       svd_chain_core_s_per_block_est=t_svd2 / 2.0 * tel.nfreq, nmodes=[int(x) for x in o2["nmodes"]])
            # the whole block through the DEVICE chain: ndof of the block (and, at m = 460, the SVD products of all 64
            # frequencies for the oracle's projections + eigh_gen)
            res = bt.svd_device(blk_d, ms=[m])
            ctx.sync()
            bs, bu, sv = (res[k][0].cpu().numpy() for k in ("beam_svd", "beam_ut", "singularvalues"))
            rec["ndof"] = int(osvd.svd_num(sv, bt.svcut)[0].sum())
            # the sigma the oracle has just computed are the GPU's yardstick (not thrown away): the two sampled frequencies
            # against the device chain on the SAME block, relative to the block's largest singular value; svnum per frequency
            smax = float(sv.max())
            sv_o = np.asarray(o2["singularvalues"])
            rec["sv_max_err_over_svmax"] = float(np.abs(sv_o - sv[fsel]).max() / smax)
            n_o = (sv_o > smax * bt.svcut).sum(axis=1)
            n_g = (sv[fsel] > smax * bt.svcut).sum(axis=1)
            rec["svnum"] = [int(x) for x in n_g]
            rec["svnum_equal"] = bool(np.array_equal(n_o, n_g))
            rec["nmodes_equal"] = bool(np.array_equal(np.asarray(o2["nmodes"]), np.asarray(res["nmodes"][0])[fsel]))
            if m == 460:
                t0 = time.perf_counter()
                cs, cn = okl.sn_covariance(bs, bu, sv, kl.signal(), kl.foreground(), kl._npower(1.0), svcut=bt.svcut)
                t1 = time.perf_counter()
                ev = okl.kl_transform_m(cs, cn)[0]
                t2 = time.perf_counter()
                rec.update(kl_projections_s=t1 - t0, kl_eigh_s=t2 - t1, kl_s=t2 - t0)
                # ... and the KL spectrum of the block against the device's (oracle fed with the device's SVD products)
                bt._dev[m] = dict(beam_svd=res["beam_svd"][0], beam_ut=res["beam_ut"][0], singularvalues=sv)
                ev_g = kl._transform_batch([m], to_host=True)[0][0]
                rec["ev_max_err_over_lambda_max"] = float(np.abs(ev_g - ev).max() / np.abs(ev).max())
                kp = ev >= kl.threshold
                rec["ev_kept_modes"] = int(kp.sum())
                rec["ev_kept_max_rel_err"] = float((np.abs(ev_g[kp] - ev[kp]) / ev[kp]).max()) if kp.any() else 0.0
                bt._dev.clear()
            del res, bs, bu
            out["samples"]["m=%d" % m] = rec
            del blk_d
        # ---- scaling (ESTIMATE): the product's cost model, BeamTransfer._m_cost = flat + linear + cubic in x = (L - m) / L
        x = lambda m_: float(L - m_) / L
        s0, s460 = out["samples"]["m=0"], out["samples"]["m=460"]
        # SVD chain: linear interpolation in x between the two measured points (x 64 frequencies).  KL: flop counts at the
        # per-core rate LAPACK's zhegvd reached on the m = 460 pencil — eig(n) = 68 n^3 / 3 and the reference's
        # projection loops 8 n^2 Lm x 16 polarisation pairs x 2 covariances (it multiplies the all-zero blocks too) — with
        # ndof(m) interpolated linearly in x between the two blocks' real ndof; never below the measured m = 460 seconds
        # (65 000 small numpy products: interpreter-bound there).
        svd_tot = sum(s460["svd_chain_core_s_per_block_est"] + (s0["svd_chain_core_s_per_block_est"] - s460["svd_chain_core_s_per_block_est"])
                      * (x(m_) - x(460)) / (x(0) - x(460)) for m_ in range(tel.mmax + 1))
        rate = (68.0 / 3.0) * s460["ndof"] ** 3 / max(s460["kl_eigh_s"], 1e-9)      # real flop / s of one core
        kl_tot, kl_m0 = 0.0, None
        for m_ in range(tel.mmax + 1):
            nd = s460["ndof"] + (s0["ndof"] - s460["ndof"]) * (x(m_) - x(460)) / (x(0) - x(460))
            nd = max(nd, 0.0)
            t_ = ((68.0 / 3.0) * nd ** 3 + 8.0 * nd * nd * (L - m_) * 16 * 2) / rate
            t_ = max(t_, s460["kl_s"] * min(1.0, (nd / max(s460["ndof"], 1)) ** 2))
            kl_tot += t_
            if m_ == 0:
                kl_m0 = t_
        out["sv_max_err_over_svmax"] = max(r["sv_max_err_over_svmax"] for r in out["samples"].values())
        out["svnum_equal"] = all(r["svnum_equal"] for r in out["samples"].values())
        out["seconds"] = sum(r["svd_chain_s_for_2_frequencies"] + r.get("kl_s", 0.0) for r in out["samples"].values())
        out["estimate"] = dict(
            svd_core_s_whole_job=svd_tot, kl_core_s_whole_job=kl_tot, kl_core_s_block_m0_est=kl_m0,
            zhegvd_rate_gflops_one_core=rate / 1e9,
            btgen="not sampled (the oracle's numpy SHT at nside 512 takes minutes per column)",
            svd_plus_kl_core_hours=(svd_tot + kl_tot) / 3600.0,
            m_blocks_per_s_on_host_cores=(tel.mmax + 1) / ((svd_tot + kl_tot) / max(out["host_cores"], 1)),
            label="ESTIMATE, not a measurement: SVD core-seconds interpolated linearly in (L - m) / L between the two sampled m (x 64 "
                  "frequencies); KL core-seconds = (68 n^3 / 3 + 256 n^2 Lm) flop at the per-core rate scipy's eigh(S, N) reached on "
                  "the m = 460 pencil, ndof interpolated between the two blocks' real ndof; perfect scaling over the host's cores "
                  "assumed; BT-gen excluded")
    with open(args.out, "w") as fh:
        json.dump(out, fh, indent=1)
    print(json.dumps(out, indent=1))


if __name__ == "__main__":
    main()
