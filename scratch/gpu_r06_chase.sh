#!/bin/bash
# by-position chase against the sweep-owning pairs: configs[1] step + the chase class
cd "$GRAFT_REPO_ROOT"; mkdir -p gpurun_out; export TMPDIR=/tmp
for mode in pos pairs; do
  DM_SB_CHASE=$mode DRIFT_BENCH_DETAIL=gpurun_out/r06b_detail_$mode.json timeout -k 10 300 python bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-north-star > gpurun_out/r06b_bench_$mode.out 2> gpurun_out/r06b_bench_$mode.err || exit 3
  python - <<P
import json
d=json.load(open("gpurun_out/r06b_detail_$mode.json"))
print("$mode", "value %.1f ms %.2f" % (d["value"], d["ms_per_step"]), {k: round(v,2) for k,v in d["kernels_ms"].items()})
P
done
