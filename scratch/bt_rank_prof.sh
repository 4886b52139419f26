#!/bin/bash
# rocprofv3 kernel stats of one configs[2] rank call of BT-gen (development aid); $1 = tag
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
rm -rf gpurun_out/btprof
timeout -k 10 500 rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/btprof -o bt -- python3 scratch/btgen_bench.py --config 3 --ranges 0:64 --skip-old --out gpurun_out/bt_prof_$1.json > gpurun_out/btprof.log 2>&1
f=$(find gpurun_out/btprof -name "*kernel_stats.csv" | head -1)
python3 - "$f" <<'PY'
import csv, sys
for i, r in enumerate(csv.DictReader(open(sys.argv[1]))):
    if i >= 12: break
    print("%-58s %4s %9.1f ms %6s%%" % (r["Name"].replace("(anonymous namespace)::", "").replace("void ", "")[:58], r["Calls"], float(r["TotalDurationNs"]) / 1e6, r["Percentage"]))
PY
rm -rf gpurun_out/btprof
tail -1 gpurun_out/btprof.log
