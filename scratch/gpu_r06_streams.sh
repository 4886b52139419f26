#!/bin/bash
# m-groups on their own HIP streams (worker threads): configs[1] step at 1 / 2 groups, dealt round-robin or in contiguous cost halves
cd "$GRAFT_REPO_ROOT"; mkdir -p gpurun_out; export TMPDIR=/tmp
for cfg in "1 rr" "2 rr" "2 contig" "3 rr"; do
  set -- $cfg
  DRIFT_BENCH_SPLIT=$2 DRIFT_BENCH_DETAIL=gpurun_out/r06d_detail_s$1$2.json timeout -k 10 300 python bench.py --steps 10 --warmup 3 --streams $1 --no-cpu-baseline --no-north-star > gpurun_out/r06d_bench_s$1$2.out 2> gpurun_out/r06d_bench_s$1$2.err || { tail -5 gpurun_out/r06d_bench_s$1$2.err; exit 3; }
  python - <<P
import json
d=json.load(open("gpurun_out/r06d_detail_s$1$2.json"))
print("streams $1 $2", "value %.1f ms %.2f" % (d["value"], d["ms_per_step"]), d["stage_ms"])
P
done
