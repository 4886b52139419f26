#!/bin/bash
# Round-3 profile set for profiles/ (tag $1): rocprofv3 kernel stats of the bench, PMC passes (FETCH_SIZE, WRITE_SIZE,
# MFMA busy) each in a pass of its own, the default bench line, the two-rank rehearsals, BT-gen kernel stats of a
# configs[2] rank call.  Every step under its own timeout; stops at the first failure.
set -o pipefail
tag=${1:-r03x}
cd "$GRAFT_REPO_ROOT"; mkdir -p gpurun_out; export TMPDIR=/tmp
TAG=$tag timeout -k 10 300 bash scratch/run_rocprof.sh > gpurun_out/prof_$tag.txt 2>&1 || exit 2
timeout -k 10 300 bash scratch/run_pmc.sh fetch_$tag FETCH_SIZE > /dev/null 2>&1 || exit 3
timeout -k 10 300 bash scratch/run_pmc.sh write_$tag WRITE_SIZE > /dev/null 2>&1 || exit 4
timeout -k 10 300 bash scratch/run_pmc.sh mfma_$tag SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES GRBM_GUI_ACTIVE > /dev/null 2>&1 || exit 5
python scratch/make_traffic_json.py $tag > gpurun_out/${tag}_traffic.txt 2>&1
python scratch/make_mfma_json.py $tag > gpurun_out/${tag}_mfma.txt 2>&1
cp profiles/${tag}_pmc_traffic.json profiles/${tag}_pmc_mfma.json gpurun_out/ 2>/dev/null
f=$(find gpurun_out/prof_$tag -name "*kernel_stats.csv" | head -1); cp "$f" gpurun_out/${tag}_bench_kernel_stats.csv
for p in fetch write mfma; do f=$(find gpurun_out/pmc_${p}_$tag -name "*counter_collection.csv" | head -1); python - "$f" gpurun_out/${tag}_pmc_${p}_by_kernel.csv <<'PY'
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
find gpurun_out/pmc_fetch_$tag gpurun_out/pmc_write_$tag gpurun_out/pmc_mfma_$tag gpurun_out/prof_$tag -name "*.csv" -size +5M -delete
timeout -k 10 600 python bench.py > gpurun_out/${tag}_bench_default.json 2> gpurun_out/${tag}_bench_default.err || exit 6
timeout -k 10 300 python bench.py --gpus 2 --one-gpu --backend gloo --no-cpu-baseline > gpurun_out/${tag}_bench2s.json 2> gpurun_out/${tag}_bench2s.err || exit 7
timeout -k 10 300 python bench.py --gpus 2 --one-gpu --backend gloo --mode weak --no-cpu-baseline > gpurun_out/${tag}_bench2w.json 2> gpurun_out/${tag}_bench2w.err || exit 8
# BT-gen of a configs[2] rank call (65 blocks) under the kernel trace
rm -rf gpurun_out/btprof
timeout -k 10 500 rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/btprof -o bt -- python3 scratch/btgen_bench.py --config 3 --ranges 0:64 --skip-old --out gpurun_out/${tag}_btgen_config3_rank.json > gpurun_out/btprof.log 2>&1 || exit 9
f=$(find gpurun_out/btprof -name "*kernel_stats.csv" | head -1); head -12 "$f" > gpurun_out/${tag}_btgen_config3_rank_kernel_stats.csv
rm -rf gpurun_out/btprof
python - $tag <<'PY'
import json, sys
tag = sys.argv[1]
for f in ("bench_default", "bench2s", "bench2w"):
    d = json.loads(open("gpurun_out/%s_%s.json" % (tag, f)).read().strip().splitlines()[-1])
    print(f, round(d["value"], 1), d["n_gpus"], round(d["ms_per_step"], 1), d["scaling"], {k: round(v, 1) for k, v in d["stage_ms"].items()},
          d["roofline"]["kernel"], round(d["roofline"]["frac"], 3))
PY
