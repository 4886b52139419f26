#!/bin/bash
# SQ counters of the Jacobi sweep kernels on a configs[2] SVD batch (scratch/svd_phase_probe.py), dispatches restricted by regex
cd /tmp && export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
for k in jac_apply_kernel jac_gram_kernel; do
  out=gpurun_out/pmc_jac_$k; rm -rf $out; mkdir -p $out
  rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_WAVE_CYCLES GRBM_GUI_ACTIVE SQ_WAIT_INST_LDS SQ_ACTIVE_INST_VALU --kernel-include-regex "$k" --kernel-trace --output-format csv -d $out -o p -- python3 scratch/svd_phase_probe.py --m0 0 --n 6 --no-debug > $out/stdout.txt 2> $out/stderr.txt
  echo "$k rc $?"
  python3 - $out <<'PY'
import csv, glob, sys
from collections import defaultdict
f = glob.glob(sys.argv[1] + "/**/*counter_collection.csv", recursive=True)[0]
v = defaultdict(float); n = set()
for r in csv.DictReader(open(f)):
    v[r["Counter_Name"]] += float(r["Counter_Value"]); n.add(r["Dispatch_Id"])
act = v["GRBM_GUI_ACTIVE"] / 8.0
print("dispatches", len(n), {k: "%.3g" % x for k, x in v.items()})
print("mfma_busy %.3f  of wave cycles: wait_any %.3f wait_inst_any %.3f active_inst_any %.3f wait_lds %.3f  waves/SIMD avg %.2f" % (
    v["SQ_VALU_MFMA_BUSY_CYCLES"] / (act * 1024), v["SQ_WAIT_ANY"] / v["SQ_WAVE_CYCLES"], v["SQ_WAIT_INST_ANY"] / v["SQ_WAVE_CYCLES"],
    v["SQ_ACTIVE_INST_ANY"] / v["SQ_WAVE_CYCLES"], v["SQ_WAIT_INST_LDS"] / v["SQ_WAVE_CYCLES"], v["SQ_WAVE_CYCLES"] / (act * 1024)))
PY
done
