#!/bin/bash
# per-rank step times of the sharded configs[1] job, each rank's m-range alone on the GPU: the expected strong-scaling curve
cd "$GRAFT_REPO_ROOT"
out=gpurun_out/${1:-r04}_shard_curve.txt
: > $out
for n in 2 4 8; do
  for r in $(seq 0 $((n-1))); do
    python bench.py --shard $r/$n --no-cpu-baseline --no-north-star --prime-passes 4 > gpurun_out/shard.json 2>/dev/null || exit 1
    python - $r $n >> $out <<'PY'
import json, sys
l = json.loads(open("gpurun_out/shard.json").read().strip().splitlines()[-1])
r = l["ranks"]["per_rank"][0]
print("N %s rank %s m %d..%d blocks %d step_ms %.2f" % (sys.argv[2], sys.argv[1], r["m_lo"], r["m_hi"], r["m_hi"] - r["m_lo"] + 1, l["ms_per_step"]))
PY
  done
done
cat $out
python - $out <<'PY'
import sys, collections
worst = collections.defaultdict(float)
for line in open(sys.argv[1]):
    w = line.split()
    worst[int(w[1])] = max(worst[int(w[1])], float(w[-1]))
for n, t in sorted(worst.items()):
    print("N = %d: slowest rank %.1f ms per step -> %.0f m-blocks/s expected (129 blocks per step), %.2f x one GPU at 133 ms" % (n, t, 129e3 / t, 133.0 / t))
PY
