"""profiles/<tag>_configs2_pmc_mfma.json from the two rocprofv3 --pmc passes of scratch/pmc_share.sh (share 0/8 of the
configs[2] job, dispatches restricted by --kernel-include-regex): MFMA-pipe utilisation of the covariance-projection
kernel (gathered-B grouped ZGEMM) and of jac_gram at the north-star workload.

  util = SQ_VALU_MFMA_BUSY_CYCLES / (GRBM_GUI_ACTIVE / 8 XCDs x 1024 SIMDs), summed over the kernel's dispatches
"""
import csv, glob, json, os, re, sys
from collections import defaultdict

tag = sys.argv[1]
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench

out = {"_note": "bench.py --workload configs2 --share 0/8 under rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE "
                "SQ_BUSY_CYCLES --kernel-include-regex <kernel> (one pass per kernel; only those dispatches are serialised); "
                "mfma_busy = SQ_VALU_MFMA_BUSY_CYCLES / (GRBM_GUI_ACTIVE / 8 x 1024 SIMDs) summed over the dispatches",
       "_build_id": bench.build_id()}
for name, pas in (("zgemm_cov", "cov"), ("jac_gram", "gram")):
    fl = glob.glob(os.path.join(ROOT, "gpurun_out", "pmc_share_%s_%s" % (tag, pas), "**", "*counter_collection.csv"), recursive=True)
    if not fl:
        out[name] = dict(error="no counter file")
        continue
    val, calls, kern = defaultdict(float), set(), set()
    for r in csv.DictReader(open(fl[0])):
        val[r["Counter_Name"]] += float(r["Counter_Value"])
        calls.add(r["Dispatch_Id"])
        kern.add(re.sub(r"\(anonymous namespace\)::|void ", "", r["Kernel_Name"]).split("(")[0])
    busy, act = val.get("SQ_VALU_MFMA_BUSY_CYCLES", 0.0), val.get("GRBM_GUI_ACTIVE", 0.0)
    out[name] = dict(kernels=sorted(kern), dispatches=len(calls), mfma_busy_cycles=busy, grbm_gui_active=act,
                     sq_busy_cycles=val.get("SQ_BUSY_CYCLES", 0.0), mfma_busy=(busy / (act / 8.0 * 1024.0)) if act > 0 else None)
    try:   # the share's own line under the profiler (wall time with the serialised dispatches)
        so = os.path.join(ROOT, "gpurun_out", "pmc_share_%s_%s" % (tag, pas), "stdout.txt")
        line = [l for l in open(so).read().splitlines() if l.startswith("{")]
        d = json.loads(line[-1])
        out[name]["share_s_under_profiler"] = d["share_s"]
        cls = d["classes"].get("zgemm_cov" if name == "zgemm_cov" else "jac_gram")
        if cls:
            out[name]["hip_event_tflops_same_run"] = cls["rate"]
    except Exception as e:
        out[name]["line_error"] = repr(e)
json.dump(out, open(os.path.join(ROOT, "profiles", "%s_configs2_pmc_mfma.json" % tag), "w"), indent=1)
print(json.dumps(out, indent=1))
