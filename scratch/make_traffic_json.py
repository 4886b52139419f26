"""Per-launch HBM traffic of the main kernels from the two PMC passes (FETCH_SIZE, WRITE_SIZE).
Corrections as MI355X_MICROARCH.md prescribes: both counters are in KiB; on gfx950 FETCH_SIZE
reports half of the bytes of wide (16 B/lane) coalesced reads -> doubled; WRITE_SIZE is exact."""
import csv, glob, json, re, sys
from collections import defaultdict
tag = sys.argv[1]
def load(d, name):
    f = glob.glob(d + "/**/*counter_collection.csv", recursive=True)[0]
    tot = defaultdict(float); calls = defaultdict(set)
    for r in csv.DictReader(open(f)):
        if r["Counter_Name"] != name: continue
        k = re.sub(r"dm_trd\d+::", "", r["Kernel_Name"].replace("(anonymous namespace)::", "").replace("void ", "")).split("(")[0]
        tot[k] += float(r["Counter_Value"]); calls[k].add(r["Dispatch_Id"])
    return tot, {k: len(v) for k, v in calls.items()}
ft, fc = load("gpurun_out/pmc_fetch_" + tag, "FETCH_SIZE")
wt, wc = load("gpurun_out/pmc_write_" + tag, "WRITE_SIZE")
out = {"_note": "bench.py --steps 1 --warmup 1 under rocprofv3 --pmc (2 passes of the step per run); bytes per launch; "
                "FETCH_SIZE KiB x 1024 x 2 (gfx950 wide-load correction), WRITE_SIZE KiB x 1024"}
top = sorted(ft, key=lambda k: -(ft[k] + wt.get(k, 0)))
for k in top[:12] + [k for k in top[12:] if "chase" in k or "q2_apply" in k or "panel_fused" in k]:   # (the chase always: its traffic is a tracked figure)
    n = max(fc.get(k, 1), 1)
    out[k] = {"launches": n, "fetch_bytes_per_launch": ft[k] * 1024 * 2 / n, "write_bytes_per_launch": wt.get(k, 0.0) * 1024 / max(wc.get(k, 1), 1)}
import os, sys as _sys
_sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench as _bench
out["_build_id"] = _bench.build_id()
json.dump(out, open("profiles/%s_pmc_traffic.json" % tag, "w"), indent=1)
print(json.dumps(out, indent=1)[:1500])
