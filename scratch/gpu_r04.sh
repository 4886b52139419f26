#!/bin/bash
# Round-4 profile set for profiles/ (tag $1): rocprofv3 kernel stats of the bench, PMC passes (FETCH_SIZE, WRITE_SIZE,
# MFMA busy) each in a pass of its own, the default bench line (with the north-star leg), the multi-rank rehearsals on one
# card, kernel stats + MFMA-busy counters of a full configs[2] share (0/8) through generate(), BT-gen kernel stats of a
# configs[2] rank call at iter 0 and iter 3.  Every step under its own timeout; stops at the first failure.
set -o pipefail
tag=${1:-r04x}
part=${2:-all}     # "a": bench profiles + rehearsals, "b": kernel stats of a configs[2] share, "c": its MFMA counters, "d": BT-gen, "all"
cd "$GRAFT_REPO_ROOT"; mkdir -p gpurun_out; export TMPDIR=/tmp
if [ "$part" = "a" ] || [ "$part" = "all" ]; then
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
echo "$(date +%T) bench default"
timeout -k 10 900 python bench.py > gpurun_out/${tag}_bench_default.json 2> gpurun_out/${tag}_bench_default.err || exit 6
echo "$(date +%T) rehearsals"
timeout -k 10 300 python bench.py --gpus 2 --one-gpu --backend gloo --no-cpu-baseline > gpurun_out/${tag}_bench_2ranks_one_gpu_gloo_sharded.json 2> gpurun_out/${tag}_bench2s.err || exit 7
timeout -k 10 300 python bench.py --gpus 2 --one-gpu --backend gloo --mode weak --no-cpu-baseline > gpurun_out/${tag}_bench_2ranks_one_gpu_gloo_weak.json 2> gpurun_out/${tag}_bench2w.err || exit 8
DRIFTMI_WORKSPACE_GB=8 timeout -k 10 400 python bench.py --gpus 6 --one-gpu --backend gloo --no-cpu-baseline --steps 3 --prime-passes 2 > gpurun_out/${tag}_bench_6ranks_one_gpu_gloo_sharded.json 2> gpurun_out/${tag}_bench6s.err || exit 9
fi
if [ "$part" = "a" ]; then exit 0; fi
# a full configs[2] share (0/8) through generate(): kernel stats, then MFMA-busy counters in a pass of their own
if [ "$part" = "b" ] || [ "$part" = "all" ]; then
echo "$(date +%T) share kernel stats"
rm -rf gpurun_out/sharetrace
timeout -k 10 900 rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/sharetrace -o sh -- python3 bench.py --workload configs2 --share 0/8 > gpurun_out/${tag}_configs2_share0of8_generate.json 2> gpurun_out/share.err || exit 10
f=$(find gpurun_out/sharetrace -name "*kernel_stats.csv" | head -1); cp "$f" gpurun_out/${tag}_configs2_share_kernel_stats.csv
rm -rf gpurun_out/sharetrace
fi
if [ "$part" = "b" ]; then exit 0; fi
if [ "$part" = "c" ] || [ "$part" = "all" ]; then
echo "$(date +%T) share pmc"
rm -rf gpurun_out/sharepmc
# (round 4: NOT completed on this pool — the counter collection segfaulted inside the HIP runtime on share 0/8, and on share 0/40
# it had not finished after 17 minutes (the counters serialise the dispatches); the covariance kernel's MFMA-busy figure comes from
# the configs[1] pass, profiles/*_pmc_mfma.json)
( while true; do sleep 45; date +%T >> gpurun_out/sharepmc_heartbeat.txt; done ) &
hb=$!
timeout -k 10 1050 rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES GRBM_GUI_ACTIVE --kernel-trace --output-format csv -d gpurun_out/sharepmc -o sh -- python3 bench.py --workload configs2 --share ${PMC_SHARE:-0/40} > gpurun_out/sharepmc.json 2> gpurun_out/sharepmc.err
rc=$?
kill $hb
[ $rc -eq 0 ] || exit 11
f=$(find gpurun_out/sharepmc -name "*counter_collection.csv" | head -1)
python - "$f" gpurun_out/${tag}_configs2_share_pmc_mfma.json $tag <<'PY'
import csv, sys, re, json, os
from collections import defaultdict
agg = defaultdict(lambda: defaultdict(float)); n = defaultdict(set)
for r in csv.DictReader(open(sys.argv[1])):
    k = re.sub(r"dm_trd\d+::", "", r["Kernel_Name"].replace("(anonymous namespace)::", "").replace("void ", "")).split("(")[0]
    agg[k][r["Counter_Name"]] += float(r["Counter_Value"]); n[k].add(r["Dispatch_Id"])
out = {"_note": "bench.py --workload configs2 --share 0/40 (seven m-blocks, m = 0..6; the counter collection of rocprofv3 segfaults inside the HIP runtime on a full share 0/8) under rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES GRBM_GUI_ACTIVE "
                "(a pass of its own); mfma_busy = SQ_VALU_MFMA_BUSY_CYCLES / (GRBM_GUI_ACTIVE / 8 x 1024 SIMDs), as profiles/*_pmc_mfma.json"}
for k, v in sorted(agg.items(), key=lambda kv: -kv[1].get("SQ_VALU_MFMA_BUSY_CYCLES", 0.0))[:14]:
    g = v.get("GRBM_GUI_ACTIVE", 0.0)
    out[k] = {"dispatches": len(n[k]), "mfma_busy": (v.get("SQ_VALU_MFMA_BUSY_CYCLES", 0.0) / (g / 8.0 * 1024.0)) if g else None,
              "SQ_VALU_MFMA_BUSY_CYCLES": v.get("SQ_VALU_MFMA_BUSY_CYCLES", 0.0), "GRBM_GUI_ACTIVE": g}
sys.path.insert(0, os.getcwd())
import bench
out["_build_id"] = bench.build_id()
json.dump(out, open(sys.argv[2], "w"), indent=1)
print(json.dumps({k: v for k, v in list(out.items())[:8]}, indent=1)[:1500])
PY
rm -rf gpurun_out/sharepmc
fi
if [ "$part" = "c" ]; then exit 0; fi
# BT-gen of a configs[2] rank call under the kernel trace, iter 0 and iter 3
echo "$(date +%T) btgen"
bash scratch/bt_iter_prof.sh ${tag}_rank0_iter3 0:32 3 > gpurun_out/${tag}_btgen_rank0_iter3.txt 2>&1 || exit 12
bash scratch/bt_iter_prof.sh ${tag}_rank0_iter0 0:32 0 > gpurun_out/${tag}_btgen_rank0_iter0.txt 2>&1 || exit 13
python scratch/bt_iter_bench.py --config 2 --ranges all --out gpurun_out/${tag}_btgen_iter_configs1.json > /dev/null 2>&1
python scratch/bt_iter_bench.py --config 3 --ranges 0:32 157:212 --reps 2 --out gpurun_out/${tag}_btgen_iter_configs2.json > /dev/null 2>&1
python - $tag <<'PY'
import json, sys
tag = sys.argv[1]
import os
for f in ("bench_default", "bench_2ranks_one_gpu_gloo_sharded", "bench_2ranks_one_gpu_gloo_weak", "bench_6ranks_one_gpu_gloo_sharded"):
    if not os.path.exists("gpurun_out/%s_%s.json" % (tag, f)):
        continue
    d = json.loads(open("gpurun_out/%s_%s.json" % (tag, f)).read().strip().splitlines()[-1])
    print(f, round(d["value"], 1), d["n_gpus"], round(d["ms_per_step"], 1), d["scaling"], {k: round(v, 1) for k, v in d["stage_ms"].items()},
          d["roofline"]["kernel"], round(d["roofline"]["frac"], 3))
    if "north_star" in d:
        ns = d["north_star"]
        print("   north_star", {k: ns.get(k) for k in ("share_s", "kernel_coverage_of_wall", "leg_wall_s")}, (ns.get("zgemm_cov") or {}).get("frac"))
PY
