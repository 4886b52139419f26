#!/bin/bash
# GPU-timeline gaps and kernel totals of one configs[2] share through generate(); $1 = share (r/N), $2 = tag
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
rm -rf gpurun_out/sharetrace
timeout -k 10 900 rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/sharetrace -o sh -- python3 bench.py --workload configs2 --share $1 > gpurun_out/share_$2.json 2> gpurun_out/share_$2.err
f=$(find gpurun_out/sharetrace -name "*kernel_trace.csv" | head -1)
g=$(find gpurun_out/sharetrace -name "*kernel_stats.csv" | head -1)
cp "$g" gpurun_out/share_$2_kernel_stats.csv
python3 - "$f" <<'PY'
import csv, sys, re
rows = [(int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"]) for r in csv.DictReader(open(sys.argv[1]))]
rows.sort()
def short(k):
    k = re.sub(r"dm_trd\d+::", "", k.replace("(anonymous namespace)::", "").replace("void ", ""))
    return k.split("(")[0][:40]
t0 = rows[0][0]; cur = rows[0][0]; busy = 0; gaps = []
for i, (s, e, k) in enumerate(rows):
    if s > cur:
        gaps.append(((s - cur) / 1e6, (s - t0) / 1e9, short(rows[i - 1][2]), short(k)))
    busy += max(0, e - max(s, cur)); cur = max(cur, e)
span = (cur - t0) / 1e9
print("span %.2f s busy %.2f s idle %.2f s, %d dispatches" % (span, busy / 1e9, span - busy / 1e9, len(rows)))
big = sorted(gaps, reverse=True)[:25]
print("gaps > 20 ms: %d totalling %.2f s; gaps 1-20 ms: %.2f s; gaps < 1 ms: %.2f s" % (
    sum(1 for g in gaps if g[0] > 20), sum(g[0] for g in gaps if g[0] > 20) / 1e3,
    sum(g[0] for g in gaps if 1 < g[0] <= 20) / 1e3, sum(g[0] for g in gaps if g[0] <= 1) / 1e3))
for g in sorted(big, key=lambda x: x[1]):
    print("t=%7.2f s gap %8.1f ms | %s -> %s" % (g[1], g[0], g[2], g[3]))
PY
python3 - gpurun_out/share_$2_kernel_stats.csv <<'PY'
import csv, sys
rows = list(csv.DictReader(open(sys.argv[1])))
tot = sum(float(r["TotalDurationNs"]) for r in rows)
print("kernel sum %.2f s" % (tot / 1e9))
for r in rows[:28]:
    print("%-80s %7s %9.1f ms %5.1f%%" % (r["Name"].replace("(anonymous namespace)::", "").replace("void ", "")[:80], r["Calls"], float(r["TotalDurationNs"]) / 1e6, 100 * float(r["TotalDurationNs"]) / tot))
PY
rm -rf gpurun_out/sharetrace
python3 -c "
import json
l = json.loads(open('gpurun_out/share_$2.json').read().strip().splitlines()[-1])
print('share_s', l['share_s'], 'kernel_s', l['kernel_s'])
"
