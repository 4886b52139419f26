#!/usr/bin/env python3
"""Where the SVD chain of a configs[2] batch spends its time: BT-gen of m = m0 .. m0 + n - 1, then `svd_device` twice —
once plain (wall + kernel classes), once under DM_DEBUG=1 (the library prints the wall time of every phase of every
dm_jacobi_rows call, synchronising at each mark).

    python scratch/svd_phase_probe.py --m0 0 --n 11 > gpurun_out/svd_phase.log 2>&1
"""
import argparse
import json
import os
import sys
import tempfile
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--m0", type=int, default=0)
    ap.add_argument("--n", type=int, default=11)
    ap.add_argument("--no-debug", action="store_true")
    ap.add_argument("--reps", type=int, default=1)
    args = ap.parse_args()
    import torch

    from driftscan_amd import beamtransfer, btgen, cylinder, device

    ctx = device.get_context(workspace_bytes=100 << 30)
    tel = cylinder.PolarisedCylinderTelescope.from_config(dict(bench.CFG3))
    with tempfile.TemporaryDirectory() as tmp:
        bt = beamtransfer.BeamTransfer(tmp, telescope=tel)
        t0 = time.perf_counter()
        beam = btgen.beam_m_all(tel, ctx=ctx, max_bytes=48 << 30, m_range=(args.m0, args.m0 + args.n - 1))
        ctx.sync()
        print("BT-gen %d blocks: %.2f s" % (args.n, time.perf_counter() - t0), flush=True)
        ms = None if os.environ.get("PROBE_NO_MS") else list(range(args.m0, args.m0 + args.n))
        res = bt.svd_device(beam, ms=ms)   # allocator warm-up
        ctx.sync()
        del res
        for rep in range(args.reps):
            ctx.prof_reset(2)
            t0 = time.perf_counter()
            res = bt.svd_device(beam, ms=ms)
            ctx.sync()
            dt = time.perf_counter() - t0
            pr = ctx.prof_report()
            sv = res["singularvalues"].cpu().numpy()
            if os.environ.get("PROBE_SAVE"):
                import numpy as _np
                _np.savez(os.environ["PROBE_SAVE"], sv=sv, nmodes=_np.asarray(res["nmodes"]))
            print(json.dumps(dict(svd_s=dt, per_block_s=dt / args.n, sweeps=getattr(ctx, "last_sweeps", None),
                                  classes_ms={k: round(v["ms"], 1) for k, v in sorted(pr.items(), key=lambda kv: -kv[1]["ms"])},
                                  sv_sum=float(sv.sum()), nmodes_mean=float((sv > 0).sum(axis=-1).mean()))), flush=True)
            del res
        if not args.no_debug:
            ctx.prof_reset(0)
            os.environ["DM_DEBUG"] = "1"
            t0 = time.perf_counter()
            res = bt.svd_device(beam, ms=ms)
            ctx.sync()
            print("with DM_DEBUG: %.2f s" % (time.perf_counter() - t0), flush=True)


if __name__ == "__main__":
    main()
