#!/bin/bash
# Kernel trace of an SVD batch (scratch/svd_phase_probe.py) -> idle gaps of the LAST chain call by (kernel before -> after)
cd /tmp && export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
rm -rf gpurun_out/gaptrace; mkdir -p gpurun_out/gaptrace
rocprofv3 --kernel-trace --output-format csv -d gpurun_out/gaptrace -o t -- python3 scratch/svd_phase_probe.py --m0 ${1:-0} --n ${2:-10} --no-debug > gpurun_out/gaptrace/stdout.txt 2> gpurun_out/gaptrace/stderr.txt
f=$(find gpurun_out/gaptrace -name "*kernel_trace.csv" | head -1)
python3 - "$f" <<'PY' > gpurun_out/gap_analysis_probe.txt
import csv, sys
from collections import defaultdict
rows = list(csv.DictReader(open(sys.argv[1])))
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
# the probe runs the chain twice (warm-up + timed): take the dispatches after the last svd_build_z_kernel ... simpler: the second half
idx = [i for i, r in enumerate(rows) if "svd_build_z_kernel" in r["Kernel_Name"]]
start = idx[len(idx) // 2] if idx else len(rows) // 2
last = rows[start:]
busy = 0; gaps = defaultdict(lambda: [0, 0.0]); end_prev = None; prev = None; hist = defaultdict(float)
for r in last:
    s, e = int(r["Start_Timestamp"]), int(r["End_Timestamp"])
    if end_prev is not None and s > end_prev:
        g = s - end_prev
        key = prev[:48] + " -> " + r["Kernel_Name"][:48]
        gaps[key][0] += 1; gaps[key][1] += g
        b = "<10us" if g < 1e4 else "<50us" if g < 5e4 else "<200us" if g < 2e5 else "<1ms" if g < 1e6 else ">=1ms"
        hist[b] += g
    end_prev = e if end_prev is None else max(end_prev, e)
    busy += e - s; prev = r["Kernel_Name"].replace("(anonymous namespace)::", "").replace("void ", "")
span = int(last[-1]["End_Timestamp"]) - int(last[0]["Start_Timestamp"])
print("dispatches %d span %.1f ms kernel %.1f ms idle %.1f ms" % (len(last), span / 1e6, busy / 1e6, sum(v[1] for v in gaps.values()) / 1e6))
print("idle by gap length (ms):", {k: round(v / 1e6, 1) for k, v in hist.items()})
for k, (c, t) in sorted(gaps.items(), key=lambda kv: -kv[1][1])[:30]:
    print("%8.1f ms %5d gaps avg %7.1f us  %s" % (t / 1e6, c, t / c / 1e3, k))
PY
head -40 gpurun_out/gap_analysis_probe.txt
rm -rf gpurun_out/gaptrace
