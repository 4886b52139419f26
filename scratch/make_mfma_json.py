"""MFMA-pipe utilisation per kernel from the rocprofv3 --pmc pass `SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES
GRBM_GUI_ACTIVE` (scratch/run_profiles_r02.sh): profiles/<tag>_pmc_mfma.json, read by bench.py (`roofline.mfma_busy`).

  busy   = SQ_VALU_MFMA_BUSY_CYCLES summed over the dispatches of the kernel (cycles in which a SIMD's matrix pipe is
           busy, summed over all SIMDs of the chip)
  window = GRBM_GUI_ACTIVE / 8 XCDs (MI355X_MICROARCH.md: rocprofv3 reports the sum over the 8 XCDs) x 1024 SIMDs
  util   = busy / window
"""
import csv, glob, json, re, sys
from collections import defaultdict

tag = sys.argv[1]
f = glob.glob("gpurun_out/pmc_mfma_%s/**/*counter_collection.csv" % tag, recursive=True)[0]
val = defaultdict(lambda: defaultdict(float))
calls = defaultdict(set)
for r in csv.DictReader(open(f)):
    k = re.sub(r"dm_trd\d+::", "", r["Kernel_Name"].replace("(anonymous namespace)::", "").replace("void ", "")).split("(")[0]
    val[k][r["Counter_Name"]] += float(r["Counter_Value"])
    calls[k].add(r["Dispatch_Id"])
out = {"_note": "bench.py --steps 1 --warmup 1 under rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES GRBM_GUI_ACTIVE; "
                "util = SQ_VALU_MFMA_BUSY_CYCLES / (GRBM_GUI_ACTIVE / 8 x 1024 SIMDs), summed over the kernel's dispatches"}
rows = []
for k, v in val.items():
    busy, act = v.get("SQ_VALU_MFMA_BUSY_CYCLES", 0.0), v.get("GRBM_GUI_ACTIVE", 0.0)
    if busy <= 0 or act <= 0:
        continue
    rows.append((busy, k, dict(launches=len(calls[k]), mfma_busy_cycles=busy, grbm_gui_active=act,
                               sq_busy_cycles=v.get("SQ_BUSY_CYCLES", 0.0), util=busy / (act / 8.0 * 1024.0))))
for _, k, rec in sorted(rows, reverse=True)[:12]:
    out[k] = rec
import os, sys as _sys
_sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench as _bench
out["_build_id"] = _bench.build_id()
json.dump(out, open("profiles/%s_pmc_mfma.json" % tag, "w"), indent=1)
print(json.dumps(out, indent=1)[:2500])
