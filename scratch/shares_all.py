#!/usr/bin/env python3
"""ALL N shares of the BASELINE configs[2] (or configs[3]) job, one after the other on this GPU, each in a fresh child
process (`bench.py --workload configs2 --share r/N`): the record behind `north_star.projected_job_s` = the MAX over the
shares (m-blocks are independent; the N-GPU job's wall time is its slowest rank's).  This parent never touches the GPU.

    python scratch/shares_all.py --n 8 --out gpurun_out/r05_configs2_shares.json
"""
import argparse
import json
import os
import subprocess
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--n", type=int, default=8)
    ap.add_argument("--workload", default="configs2")
    ap.add_argument("--out", required=True)
    ap.add_argument("--shares", default=None, help="comma-separated ranks (default: all)")
    ap.add_argument("--files", action="store_true",
                    help="every share WRITES its product files (bitshuffle + LZ4, blocks truncated on the device: the reference's "
                         "production setting) into a temporary directory under --outdir, removed after the share")
    ap.add_argument("--outdir", default=None)
    ap.add_argument("--pause", type=float, default=10.0,
                    help="seconds between two shares: the driver wipes the ~240 GB the previous process freed in the background, and "
                         "the next process's first large allocations wait for it (2-3 s in front of its BT-gen kernels when the shares "
                         "run back to back; a rank of a real job starts on an idle card)")
    args = ap.parse_args()
    import bench

    ranks = [int(x) for x in args.shares.split(",")] if args.shares else list(range(args.n))
    shares = []
    for k, r in enumerate(ranks):
        if k and args.pause > 0:
            time.sleep(args.pause)
        t0 = time.perf_counter()
        cmd = [sys.executable, os.path.join(ROOT, "bench.py"), "--workload", args.workload, "--share", "%d/%d" % (r, args.n)]
        env = dict(os.environ)
        if args.files:
            cmd += ["--files", "--truncate"] + (["--outdir", args.outdir] if args.outdir else [])
            env["DRIFTMI_H5_CODEC"] = "bitshuffle"
        res = subprocess.run(cmd, stdout=subprocess.PIPE, stderr=subprocess.PIPE, timeout=900, env=env)
        if res.returncode != 0:
            print("share %d failed: %s" % (r, res.stderr.decode()[-800:]), file=sys.stderr, flush=True)
            shares.append(dict(share="%d/%d" % (r, args.n), error=res.stderr.decode()[-400:]))
            continue
        d = json.loads(res.stdout.decode().strip().splitlines()[-1])
        st = d.get("stages") or {}
        rec = dict(share="%d/%d" % (r, args.n), m_range=d.get("m_range"), m_blocks=int(round(d["value"] * d["share_s"])),
                   share_s=d["share_s"], kernel_s=d["kernel_s"], hbm_peak_gb=d["hbm_peak_gb"],
                   stages={k: dict(seconds=v["seconds"], frac_of_fp64_mfma_peak=v["frac_of_fp64_mfma_peak"], work_flop=v["work_flop"],
                                   classes_ms=v["classes_ms"]) for k, v in st.items() if isinstance(v, dict)},
                   zgemm_cov=None if not d.get("zgemm_cov") else dict(frac=d["zgemm_cov"]["frac"], ms=d["zgemm_cov"]["ms"]),
                   zgemm_grouped_frac=(d["classes"].get("zgemm_grouped") or {}).get("frac"),
                   child_wall_s=time.perf_counter() - t0, budgets_gb=d["config"].get("budgets_gb"))
        if args.files:
            rec.update(file_bytes=d.get("file_bytes"), other_s=st.get("other_s"), io=d.get("io"),
                       kernel_coverage_of_wall=d.get("kernel_coverage_of_wall"))
        shares.append(rec)
        print("share %d/%d: m = %s  %.2f s (btgen %.1f svd %.1f kl %.1f)" % (
            r, args.n, rec["m_range"], rec["share_s"], rec["stages"].get("btgen", {}).get("seconds", 0.0),
            rec["stages"].get("svd", {}).get("seconds", 0.0), rec["stages"].get("kl", {}).get("seconds", 0.0)), flush=True)
    ok = [s for s in shares if "share_s" in s]
    out = dict(_build_id=bench.build_id(), n=args.n, workload=args.workload, shares=shares,
               max_s=max(s["share_s"] for s in ok) if ok else None,
               slowest=max(ok, key=lambda s: s["share_s"])["share"] if ok else None,
               mean_s=sum(s["share_s"] for s in ok) / len(ok) if ok else None,
               pause_s=args.pause, files=bool(args.files), outdir=args.outdir,
               note="every share of the cost-balanced contiguous partition through ProductManager.generate() on one MI355X, "
                    "products left in HBM; the job's wall time on N GPUs is max_s")
    with open(args.out, "w") as fh:
        json.dump(out, fh, indent=1)
    print(json.dumps(dict(max_s=out["max_s"], slowest=out["slowest"], mean_s=out["mean_s"])))


if __name__ == "__main__":
    main()
