"""The ONE JSON line the driver parses, and the detail record that goes beside it.

`bench.py` measures into one large dictionary (every kernel class, every rank, the north-star shares, the notes that say
how each figure was taken).  The driver reads the LAST stdout line and gave up on it once it passed ~20 KB (round 5:
`BENCH_r05.json.parsed = null`).  So the record is split:

  * `compact(full)`  -> the driver line: the contract keys, `roofline`, `cpu_baseline`, `parity`, `north_star`, scalars
    only, never more than MAX_LINE_BYTES (tests/test_bench_line.py holds it to that on a canned full record);
  * `emit(full, path)` writes the full record to `bench_detail.json`, prints it as an EARLIER stdout line (`bench_detail {...}`:
    prefixed, not itself a JSON object line) and prints the compact line LAST.
"""
import json
import math
import os
import sys

MAX_LINE_BYTES = 8192

CONTRACT_KEYS = ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling",
                 "vs_baseline", "dtype", "data")
ROOFLINE_KEYS = ("bound", "kernel", "achieved", "peak", "unit", "frac", "traffic", "traffic_unit", "traffic_source",
                 "avg_launch_us", "launches", "flops_per_launch", "bytes_per_launch", "build_id")
CPU_KEYS = ("value", "unit", "cores", "kind", "wall_s", "extrapolated", "host_cores", "error")
PARITY_KEYS = ("green", "blocks", "sv_max_err_over_svmax", "svnum_equal", "ev_kept_max_rel_err", "ev_kept_modes",
               "ev_max_err_over_lambda_max", "ev_blocks_over_pencil_tol", "kept_counts_equal")
CONFIG_KEYS = ("workload", "nfreq", "nbase", "lmax", "mmax", "sht_iter", "mode", "ranks", "backend", "shard")


def _num(x, digits=6):
    """Floats to `digits` significant digits (the line is for reading; the detail record keeps every bit)."""
    if isinstance(x, bool) or x is None:
        return x
    if isinstance(x, float):
        if math.isnan(x) or math.isinf(x):
            return None
        if x == 0.0:
            return 0.0
        return float("%.*g" % (digits, x))
    if isinstance(x, dict):
        return {k: _num(v, digits) for k, v in x.items()}
    if isinstance(x, (list, tuple)):
        return [_num(v, digits) for v in x]
    try:   # numpy scalars
        import numpy as np

        if isinstance(x, np.generic):
            return _num(x.item(), digits)
    except Exception:
        pass
    return x


def _clip(s, n):
    if not isinstance(s, str) or len(s) <= n:
        return s
    return s[: n - 3] + "..."


def _pick(d, keys):
    return {k: d[k] for k in keys if isinstance(d, dict) and k in d}


def _north_star(ns):
    """The north-star leg in scalars: the projected job (MAX over the shares), which shares this run measured live and their
    seconds, the covariance-GEMM fraction of the fp64 MFMA peak by time and by counter; at N > 1 the REAL job's seconds."""
    if not isinstance(ns, dict):
        return None
    if "error" in ns and len(ns) <= 3:
        return {"error": _clip(str(ns["error"]), 300)}
    out = {}
    if "job" in ns or "what" in ns:   # --gpus N > 1: the real N-rank job
        job = ns.get("job") or {}
        out["job_s"] = job.get("job_s")
        out["job_m_blocks_per_s"] = job.get("value")
        out["ranks"] = job.get("n_gpus")
        for k in ("ranks_seen_by_rccl", "ranks_seen_by_gloo"):   # which backend the job's tensor collective ran over
            if k in job:
                out[k] = job[k]
        per = job.get("per_rank") or []
        out["rank_seconds"] = [p.get("seconds") for p in per]
        out["rank_collective_s"] = [p.get("collective_s") for p in per]
        out["imbalance_max_over_mean"] = job.get("imbalance_max_over_mean")
        if "error" in ns:
            out["error"] = _clip(str(ns["error"]), 300)
        out["leg_wall_s"] = ns.get("leg_wall_s")
        out["target"] = "configs[2] product set < 600 s on 8 GPUs"
        return out
    out["workload"] = _clip(ns.get("workload"), 160)
    out["projected_job_s"] = ns.get("projected_job_s")
    out["projected_job_slowest_share"] = ns.get("projected_job_slowest_share")
    out["job_m_blocks_per_s"] = ns.get("job_m_blocks_per_s")
    shares = ns.get("shares") or {}
    out["live_share_s"] = {k: v.get("share_s") for k, v in shares.items() if v.get("source") == "live"}
    out["committed_share_s"] = {k: v.get("share_s") for k, v in shares.items() if v.get("source") != "live"}
    stale = [k for k, v in shares.items() if v.get("stale")]
    if stale:
        out["committed_shares_from_another_build"] = True
    if ns.get("projected_job_with_files_s") is not None:
        out["projected_job_with_files_s"] = ns.get("projected_job_with_files_s")
    st = ns.get("stages")
    if isinstance(st, dict):
        out["share_stage_s"] = {k: v.get("seconds") for k, v in st.items() if isinstance(v, dict) and "seconds" in v}
    zc = ns.get("zgemm_cov")
    if isinstance(zc, dict):
        out["zgemm_cov"] = _pick(zc, ("frac", "tflops", "ms", "launches", "mfma_busy"))
    out["share"] = ns.get("share")
    out["share_s"] = ns.get("share_s")
    out["hbm_peak_gb"] = ns.get("hbm_peak_gb")
    out["kernel_coverage_of_wall"] = ns.get("kernel_coverage_of_wall")
    cs = ns.get("cpu_sample")
    if isinstance(cs, dict):
        out["cpu_sample"] = _pick(cs, ("sv_max_err_over_svmax", "svnum_equal", "seconds", "source", "live"))
    out["leg_wall_s"] = ns.get("leg_wall_s")
    out["target"] = "configs[2] product set < 600 s on 8 GPUs; covariance GEMMs >= 0.5 of fp64 MFMA peak"
    return out


def compact(full, detail_file="bench_detail.json"):
    """The driver line from the full record.  Deterministic, scalars only; notes shortened or dropped until the serialised
    line fits MAX_LINE_BYTES (it is ~3 KB in practice)."""
    line = {k: full.get(k) for k in CONTRACT_KEYS}
    cfg = full.get("config") or {}
    c = _pick(cfg, CONFIG_KEYS)
    c["workload"] = _clip(c.get("workload"), 260)
    line["config"] = {k: v for k, v in c.items() if v is not None}
    rf = full.get("roofline")
    if isinstance(rf, dict):
        r = {k: v for k, v in _pick(rf, ROOFLINE_KEYS).items() if v is not None or k == "traffic"}
        r["traffic_source"] = _clip(r.get("traffic_source"), 120)
        if rf.get("mfma_busy_dominant") is not None:   # counter figure of the dominant class only (table: detail record)
            r["mfma_busy"] = rf["mfma_busy_dominant"]
        line["roofline"] = r
    else:
        line["roofline"] = None
    cb = full.get("cpu_baseline")
    if isinstance(cb, dict):
        o = _pick(cb, CPU_KEYS)
        o["sample"] = _clip(cb.get("sample"), 240)
        line["cpu_baseline"] = o
    else:
        line["cpu_baseline"] = None
    pr = full.get("parity")
    line["parity"] = _pick(pr, PARITY_KEYS) if isinstance(pr, dict) else None
    if isinstance(full.get("stage_ms"), dict):
        line["stage_ms"] = full["stage_ms"]
    if isinstance(full.get("stages"), dict):
        line["stage_frac_of_fp64_mfma_peak"] = {k: v.get("frac_of_fp64_mfma_peak") for k, v in full["stages"].items()
                                                if isinstance(v, dict)}
    rk = full.get("ranks")
    if isinstance(rk, dict) and rk.get("per_rank"):
        line["rank_step_ms"] = [r.get("step_ms") for r in rk["per_rank"]]
        line["rank_m_ranges"] = [[r.get("m_lo"), r.get("m_hi")] for r in rk["per_rank"]]
    for k in ("ranks_seen_by_rccl", "ranks_seen_by_gloo", "job_s"):
        if k in full:
            line[k] = full[k]
    if "north_star" in full:
        line["north_star"] = _north_star(full["north_star"])
    line["detail"] = detail_file
    line = _num(line)
    # the guarantee: whatever a future leg adds, the last line stays below the cap
    for drop in ((), ("rank_m_ranges",), ("stage_frac_of_fp64_mfma_peak", "stage_ms"), ("north_star",)):
        for k in drop:
            line.pop(k, None)
        if len(json.dumps(line)) < MAX_LINE_BYTES:
            break
    else:
        line = {k: line.get(k) for k in CONTRACT_KEYS + ("config", "roofline", "cpu_baseline", "parity", "detail")}
        line["config"] = {"workload": _clip((line.get("config") or {}).get("workload"), 120)}
    return line


def emit(full, detail_path=None, out=None):
    """Write the full record to `detail_path` (default: bench_detail.json in the working directory, or $DRIFT_BENCH_DETAIL),
    print it as one earlier line, then the compact line — the LAST line of stdout."""
    out = out or sys.stdout
    detail_path = detail_path or os.environ.get("DRIFT_BENCH_DETAIL", "bench_detail.json")
    try:
        with open(detail_path, "w") as fh:
            json.dump(full, fh, indent=1, default=str)
        shown = detail_path
    except OSError as e:   # a read-only working directory must not cost the line
        shown = "not written (%s); see the `bench_detail` line on stdout" % e.__class__.__name__
    # (prefixed, so that the compact line is the ONLY stdout line that is a JSON object: whichever way a consumer picks
    # "the JSON line" — the last line, the first line that starts with a brace, the last line that parses — it gets that one)
    out.write("bench_detail " + json.dumps(full, default=str) + "\n")
    line = compact(full, detail_file=shown)
    s = json.dumps(line)
    assert len(s) < MAX_LINE_BYTES
    out.write(s + "\n")
    out.flush()
    return line
