#!/usr/bin/env python3
"""ProductManager.generate() on BASELINE configs[1] WITH the product files (beam_m, svd, KL modes, spectra):
the end-to-end wall time a user sees, against the compute-only step of bench.py.

    python scratch/e2e_config2.py [outdir]
"""
import json
import os
import sys
import tempfile
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

import yaml  # noqa: E402


def main():
    from bench import CFG2
    from driftscan_amd import device, manager

    out = sys.argv[1] if len(sys.argv) > 1 else tempfile.mkdtemp(prefix="e2e_cfg2_")
    conf = dict(
        config=dict(beamtransfers=True, kltransform=True, psfisher=False, output_directory=out + "/prod", truncate=False),
        telescope=dict(type="UnpolarisedCylinder", **CFG2),
        kltransform=[dict(type="KLTransform", name="kl", threshold=0.1)],
    )
    cfile = out + "/params.yaml"
    os.makedirs(out, exist_ok=True)
    open(cfile, "w").write(yaml.dump(conf))
    device.get_context(workspace_bytes=int(os.environ["E2E_WS_GB"]) << 30 if os.environ.get("E2E_WS_GB") else None)
    res = {}
    for tag in ("first", "second"):   # the second pass re-uses the allocator state of the first (regen)
        pm = manager.ProductManager.from_config(cfile)
        bt = pm.beamtransfer
        kl = pm.kltransforms["kl"]
        t0 = time.perf_counter()
        bt._generate_dirs()
        bt._generate_mfiles(regen=True)
        t1 = time.perf_counter()
        bt._generate_svdfiles(regen=True)
        t2 = time.perf_counter()
        kl.generate(regen=True)
        t3 = time.perf_counter()
        size = 0
        for root, _, files in os.walk(out + "/prod"):
            size += sum(os.path.getsize(os.path.join(root, f)) for f in files)
        res[tag] = dict(beam_m_s=t1 - t0, svd_s=t2 - t1, kl_s=t3 - t2, total_s=t3 - t0, product_gb=size / 2 ** 30)
        print(tag, json.dumps(res[tag]), flush=True)
    print(json.dumps(dict(workload="configs[1] end to end with files", **res["second"])))


if __name__ == "__main__":
    main()
