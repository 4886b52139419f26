#!/usr/bin/env python3
"""BT-gen with healpy's iter = 0 against iter = 3 (harmonic-space refinement, dm_bt_columns_iter): seconds per call.

    python scratch/bt_iter_bench.py --config 2 --ranges all --out gpurun_out/bt_iter_c1.json
    python scratch/bt_iter_bench.py --config 3 --ranges 0:34 163:217 373:512 --iters 0 3
"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from bench import CFG2, CFG3  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--config", type=int, default=2, help="2 = BASELINE configs[1] (32-feed), 3 = configs[2] (128-feed)")
    ap.add_argument("--ranges", nargs="+", default=["all"])
    ap.add_argument("--iters", nargs="+", type=int, default=[0, 3])
    ap.add_argument("--bt-gb", type=float, default=48.0)
    ap.add_argument("--reps", type=int, default=3)
    ap.add_argument("--out", default=None)
    args = ap.parse_args()
    import torch

    from driftscan_amd import btgen, cylinder, device

    cfg = {2: CFG2, 3: CFG3}[args.config]
    klass = cylinder.UnpolarisedCylinderTelescope if args.config == 2 else cylinder.PolarisedCylinderTelescope
    ctx = device.get_context(workspace_bytes=int(args.bt_gb * 1.3) << 30)
    out = dict(config=args.config, runs=[])
    for rg in args.ranges:
        rec = dict(m_range=rg)
        for it in args.iters:
            tel = klass.from_config(dict(cfg, sht_iter=it))
            mr = None if rg == "all" else tuple(int(x) for x in rg.split(":"))
            ts = []
            for rep in range(args.reps + 1):
                ctx.sync(); torch.cuda.synchronize()
                t0 = time.perf_counter()
                bm = btgen.beam_m_all(tel, ctx=ctx, max_bytes=int(args.bt_gb * (1 << 30)), m_range=mr)
                ctx.sync(); torch.cuda.synchronize()
                ts.append(time.perf_counter() - t0)
                del bm
            rec["iter%d_s" % it] = min(ts[1:])
            rec["iter%d_first_s" % it] = ts[0]
            print(time.strftime("%H:%M:%S"), "config", args.config, "m", rg, "iter", it, "%.4f s (first %.3f)" % (min(ts[1:]), ts[0]), flush=True)
        if "iter0_s" in rec and "iter3_s" in rec:
            rec["ratio"] = rec["iter3_s"] / rec["iter0_s"]
        out["runs"].append(rec)
    print(json.dumps(out))
    if args.out:
        with open(args.out, "w") as fh:
            json.dump(out, fh, indent=1)


if __name__ == "__main__":
    main()
