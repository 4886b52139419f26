#!/bin/bash
# PMC passes over the bench for the band-stage kernels (development aid)
cd $GRAFT_REPO_ROOT
timeout -k 10 300 bash scratch/run_pmc.sh q2a SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_WAVE_CYCLES SQ_BUSY_CYCLES > /dev/null 2>&1 || echo "pass a failed"
timeout -k 10 300 bash scratch/run_pmc.sh q2b SQ_WAIT_INST_LDS SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT SQ_WAIT_ANY > /dev/null 2>&1 || echo "pass b failed"
timeout -k 10 300 bash scratch/run_pmc.sh q2c SQ_INSTS_LDS SQ_WAVES SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY > /dev/null 2>&1 || echo "pass c failed"
python3 - <<'PY'
import csv, glob
from collections import defaultdict
agg = defaultdict(lambda: defaultdict(float))
for f in glob.glob("gpurun_out/pmc_q2[abc]/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        k = r["Kernel_Name"]
        for key in ("sb_q2_apply", "sb_chase2", "sb_panel_fused", "jac_inner", "zgemm4_grouped_kernel<false, false, 1>"):
            if key in k:
                agg[key][r["Counter_Name"]] += float(r["Counter_Value"])
for k, v in agg.items():
    print(k)
    for c, x in sorted(v.items()): print("   %-28s %.4g" % (c, x))
PY
find gpurun_out/pmc_q2a gpurun_out/pmc_q2b gpurun_out/pmc_q2c -name "*.csv" -size +2M -delete
