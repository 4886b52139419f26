#!/bin/bash
# rocprofv3 kernel stats of one configs[2] rank call of BT-gen at a given healpy iter; $1 = tag, $2 = m range (a:b), $3 = iter
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
rm -rf gpurun_out/btprof
timeout -k 10 600 rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/btprof -o bt -- python3 scratch/bt_iter_bench.py --config 3 --ranges $2 --iters $3 --reps 1 --out gpurun_out/bt_iter_prof_$1.json > gpurun_out/btprof_$1.log 2>&1
f=$(find gpurun_out/btprof -name "*kernel_stats.csv" | head -1)
cp "$f" gpurun_out/bt_iter_prof_$1_kernel_stats.csv
python3 - "$f" <<'PY'
import csv, sys
for i, r in enumerate(csv.DictReader(open(sys.argv[1]))):
    if i >= 16: break
    print("%-70s %5s %9.1f ms %6s%%" % (r["Name"].replace("(anonymous namespace)::", "").replace("void ", "")[:70], r["Calls"], float(r["TotalDurationNs"]) / 1e6, r["Percentage"]))
PY
rm -rf gpurun_out/btprof
tail -2 gpurun_out/btprof_$1.log
