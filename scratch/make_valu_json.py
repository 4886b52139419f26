"""Vector-ALU utilisation per kernel from a rocprofv3 --pmc pass `SQ_ACTIVE_INST_VALU SQ_BUSY_CYCLES GRBM_GUI_ACTIVE`
(scratch/run_pmc.sh valu_<tag> ...): gpurun_out/<tag>_pmc_valu.json.

  SQ_ACTIVE_INST_VALU counts quad-cycles (MI355X_MICROARCH.md, constants table) in which a SIMD issues vector instructions,
  summed over all SIMDs;  window = GRBM_GUI_ACTIVE / 8 XCDs x 1024 SIMDs;  valu_busy = 4 x SQ_ACTIVE_INST_VALU / window
  (rocprof's derived VALUBusy)."""
import csv, glob, json, os, re, sys
from collections import defaultdict

tag = sys.argv[1]
f = glob.glob("gpurun_out/pmc_valu_%s/**/*counter_collection.csv" % tag, recursive=True)[0]
val = defaultdict(lambda: defaultdict(float))
calls = defaultdict(set)
for r in csv.DictReader(open(f)):
    k = re.sub(r"dm_trd\d+::", "", r["Kernel_Name"].replace("(anonymous namespace)::", "").replace("void ", "")).split("(")[0]
    val[k][r["Counter_Name"]] += float(r["Counter_Value"])
    calls[k].add(r["Dispatch_Id"])
out = {"_note": "bench.py --steps 1 --warmup 1 under rocprofv3 --pmc SQ_ACTIVE_INST_VALU SQ_BUSY_CYCLES GRBM_GUI_ACTIVE (a pass of its own); "
                "valu_busy = 4 x SQ_ACTIVE_INST_VALU / (GRBM_GUI_ACTIVE / 8 x 1024 SIMDs), summed over the kernel's dispatches"}
rows = []
for k, v in val.items():
    a, act = v.get("SQ_ACTIVE_INST_VALU", 0.0), v.get("GRBM_GUI_ACTIVE", 0.0)
    if act <= 0:
        continue
    rows.append((act, k, dict(launches=len(calls[k]), sq_active_inst_valu=a, grbm_gui_active=act, valu_busy=4.0 * a / (act / 8.0 * 1024.0))))
for _, k, rec in sorted(rows, reverse=True)[:16]:
    out[k] = rec
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench as _bench
out["_build_id"] = _bench.build_id()
json.dump(out, open("gpurun_out/%s_pmc_valu.json" % tag, "w"), indent=1)
for k, v in out.items():
    if isinstance(v, dict):
        print("%-50s %5d  %.3f" % (k[:50], v["launches"], v["valu_busy"]))
