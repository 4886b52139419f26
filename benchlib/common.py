"""What every leg of bench.py shares: the BASELINE telescopes, the peaks the rooflines are priced against, the kernel-class
lists, the identity of the build, the host's usable cores, per-class tables and the algorithmic work per stage."""
import os

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
BENCH = os.path.join(ROOT, "bench.py")   # the legs start children of the bench itself (`--cpu-worker`, `--workload configs2 --share r/N`)

CFG2 = dict(num_freq=16, freq_start=400.0, freq_end=450.0, freq_mode="edge", num_cylinders=2, cylinder_width=5.0,
            num_feeds=16, feed_spacing=0.4, tsys=1.0, force_lmax=128, force_mmax=128,
            sht_iter=3)   # healpy's documented default, stated explicitly: the measured configuration does not move with the library's default


CFG3 = dict(num_freq=64, freq_start=400.0, freq_end=500.0, freq_mode="edge", num_cylinders=4, cylinder_width=12.0,
            num_feeds=16, feed_spacing=0.4, tsys=1.0, force_lmax=512, force_mmax=512, sht_iter=3)


CFG5 = dict(num_freq=256, freq_start=400.0, freq_end=800.0, freq_mode="edge", num_cylinders=4, cylinder_width=14.5,
            num_feeds=64, feed_spacing=0.3, tsys=1.0, force_lmax=1024, force_mmax=1024, sht_iter=3)


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
