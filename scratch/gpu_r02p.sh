#!/bin/bash
set -o pipefail
mkdir -p gpurun_out
cd "$GRAFT_REPO_ROOT"
bash scratch/run_profiles_r02.sh r02p 2>&1 | tail -40
python scratch/make_traffic_json.py r02p > gpurun_out/r02p_traffic.txt 2>&1
python scratch/make_mfma_json.py r02p > gpurun_out/r02p_mfma.txt 2>&1
cp profiles/r02p_pmc_traffic.json profiles/r02p_pmc_mfma.json gpurun_out/ 2>/dev/null
f=$(find gpurun_out/prof_r02p -name "*kernel_stats.csv" | head -1); cp "$f" gpurun_out/r02p_bench_kernel_stats.csv
for p in fetch write mfma; do f=$(find gpurun_out/pmc_${p}_r02p -name "*counter_collection.csv" | head -1); python - "$f" gpurun_out/r02p_pmc_${p}_by_kernel.csv <<'PY'
import csv, sys, re
from collections import defaultdict
agg = defaultdict(lambda: defaultdict(float)); n = defaultdict(set)
for r in csv.DictReader(open(sys.argv[1])):
    k = re.sub(r"dm_trd\d+::", "", r["Kernel_Name"].replace("(anonymous namespace)::", "").replace("void ", "")).split("(")[0]
    agg[k][r["Counter_Name"]] += float(r["Counter_Value"]); n[k].add(r["Dispatch_Id"])
names = sorted({c for v in agg.values() for c in v})
w = csv.writer(open(sys.argv[2], "w")); w.writerow(["kernel", "dispatches"] + names)
for k, v in sorted(agg.items(), key=lambda kv: -sum(kv[1].values())): w.writerow([k, len(n[k])] + [v.get(c, 0.0) for c in names])
PY
done
find gpurun_out/pmc_fetch_r02p gpurun_out/pmc_write_r02p gpurun_out/pmc_mfma_r02p gpurun_out/prof_r02p -name "*.csv" -size +5M -delete
timeout -k 10 600 python bench.py > gpurun_out/r02p_bench_default.json 2> gpurun_out/r02p_bench_default.err
tail -c 1500 gpurun_out/r02p_bench_default.json
