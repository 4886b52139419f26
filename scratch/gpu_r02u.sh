#!/bin/bash
# final round-2 pass: profiles (kernel stats, PMC traffic, MFMA busy), default bench line, two-rank rehearsal
set -o pipefail
mkdir -p gpurun_out
cd "$GRAFT_REPO_ROOT"
bash scratch/run_profiles_r02.sh r02u 2>&1 | tail -30
python scratch/make_traffic_json.py r02u > gpurun_out/r02u_traffic.txt 2>&1
python scratch/make_mfma_json.py r02u > gpurun_out/r02u_mfma.txt 2>&1
cp profiles/r02u_pmc_traffic.json profiles/r02u_pmc_mfma.json gpurun_out/ 2>/dev/null
f=$(find gpurun_out/prof_r02u -name "*kernel_stats.csv" | head -1); cp "$f" gpurun_out/r02u_bench_kernel_stats.csv
for p in fetch write mfma; do f=$(find gpurun_out/pmc_${p}_r02u -name "*counter_collection.csv" | head -1); python - "$f" gpurun_out/r02u_pmc_${p}_by_kernel.csv <<'PY'
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
find gpurun_out/pmc_fetch_r02u gpurun_out/pmc_write_r02u gpurun_out/pmc_mfma_r02u gpurun_out/prof_r02u -name "*.csv" -size +5M -delete
timeout -k 10 600 python bench.py > gpurun_out/r02u_bench_default.json 2> gpurun_out/r02u_bench_default.err
tail -c 300 gpurun_out/r02u_bench_default.json
timeout -k 10 300 python bench.py --gpus 2 --one-gpu --backend gloo --no-cpu-baseline > gpurun_out/r02u_bench2.json 2> gpurun_out/r02u_bench2.err || exit 4
timeout -k 10 300 python bench.py --gpus 2 --one-gpu --backend gloo --mode sharded --no-cpu-baseline > gpurun_out/r02u_bench2s.json 2> gpurun_out/r02u_bench2s.err || exit 5
python - <<'PY'
import json
for f in ("r02u_bench_default.json","r02u_bench2.json","r02u_bench2s.json"):
    d=json.loads(open("gpurun_out/"+f).read().strip().splitlines()[-1])
    print(f, round(d["value"],1), d["n_gpus"], round(d["ms_per_step"],1), d["scaling"], {k: round(v,1) for k,v in d["stage_ms"].items()}, round(d["roofline"]["frac"],3))
PY
