#!/usr/bin/env python3
"""BT-gen timing: fused path (dm_bt_columns: synthesis inside the ring DFT) against the two-call path
(dm_bt_maps + dm_bt_sht_range, DRIFTMI_BT_MAPS=1), and their agreement.

    python scratch/btgen_bench.py --config 3 --ranges 0:1 100:107 --out gpurun_out/btgen_bench.json
"""
import argparse
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from scratch.config3_probe import CFG3, CFG5  # noqa: E402
from bench import CFG2  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--config", type=int, default=3)
    ap.add_argument("--ranges", nargs="+", default=["0:1"])
    ap.add_argument("--bt-gb", type=float, default=48.0)
    ap.add_argument("--skip-old", action="store_true")
    ap.add_argument("--out", default="gpurun_out/btgen_bench.json")
    args = ap.parse_args()
    import torch

    from driftscan_amd import btgen, cylinder, device

    cfg = {2: CFG2, 3: CFG3, 5: CFG5}[args.config]
    klass = cylinder.UnpolarisedCylinderTelescope if args.config == 2 else cylinder.PolarisedCylinderTelescope
    tel = klass.from_config(dict(cfg))
    ctx = device.get_context(workspace_bytes=int(args.bt_gb * 1.2) << 30)
    out = dict(config=args.config, runs=[])
    for rg in args.ranges:
        a, b = [int(x) for x in rg.split(":")]
        rec = dict(m_range=[a, b])
        res = {}
        for mode in (["fused"] if args.skip_old else ["fused", "maps"]):
            os.environ["DRIFTMI_BT_MAPS"] = "1" if mode == "maps" else "0"
            for rep in range(2):
                ctx.sync(); torch.cuda.synchronize()
                t0 = time.perf_counter()
                bm = btgen.beam_m_all(tel, ctx=ctx, max_bytes=int(args.bt_gb * (1 << 30)),
                                      m_range=None if (a == 0 and b >= tel.mmax) else (a, b))
                ctx.sync(); torch.cuda.synchronize()
                dt = time.perf_counter() - t0
                if rep == 0:
                    rec[mode + "_first_s"] = dt
                    del bm            # the timed repeat must not allocate a second block array next to the first
                    torch.cuda.empty_cache()
                    continue
                rec[mode + "_s"] = dt
            res[mode] = bm
            print(time.strftime("%H:%M:%S"), "config", args.config, "m", rg, mode, "%.3f s" % rec[mode + "_s"], flush=True)
        if len(res) == 2:
            sc = res["maps"].abs().max().item()
            rec["max_abs_diff_over_scale"] = (res["fused"] - res["maps"]).abs().max().item() / sc
            print("   fused vs maps path: %.2e of the block scale" % rec["max_abs_diff_over_scale"], flush=True)
        del res
        torch.cuda.empty_cache()
        # algorithmic flops of the two MFMA kernels of the fused path: the ring transform does
        # 8 npix_above_horizon nm P real flops per (f, b) column (half of the sphere lies below the horizon: npix / 2),
        # the Legendre products W_A = 8 Nring sum_m (L - m) P per column (real x complex: 4 per multiply-add pair)
        from driftscan_amd import healpix

        bb, ff = [x.ravel() for x in np.meshgrid(np.arange(tel.nbase), np.arange(tel.nfreq), indexing="ij")]
        lm, _ = tel.baseline_lmax(bb, ff)
        P = tel.num_pol_sky
        nm = min(b, tel.mmax) - a + 1
        ring = leg = 0.0
        for l in np.unique(lm):
            ncol = int((lm == l).sum())
            nside = healpix.nside_for_lmax(int(l), tel.accuracy_boost if P == 1 else 1)
            ring += 8.0 * (6.0 * nside * nside) * nm * (2 if a > 0 or nm > 1 else 1) * ncol * P
            leg += 4.0 * (4 * nside - 1) * sum(max(int(l) + 1 - m, 0) for m in range(a, a + nm)) * 2 * ncol * P
        rec["ring_transform_flops"] = ring   # +m and -m are separate output columns of the transform
        rec["legendre_flops"] = leg
        out["runs"].append(rec)
        json.dump(out, open(args.out, "w"), indent=1)


if __name__ == "__main__":
    main()
