"""Do the kernels of two m-groups on two HIP streams actually overlap?  rocprofv3 kernel trace of bench.py --streams 2: for the
last third of the dispatches: span, union of busy intervals, sum of kernel durations, per-queue sums, and how much of the
time two or more kernels were in flight."""
import csv
import sys
from collections import defaultdict

rows = list(csv.DictReader(open(sys.argv[1])))
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
last = rows[-(len(rows) // 3):]
ev = []
perq = defaultdict(float)
tot = 0
for r in last:
    s, e = int(r["Start_Timestamp"]), int(r["End_Timestamp"])
    ev.append((s, 1)); ev.append((e, -1))
    perq[r.get("Queue_Id", "?")] += e - s
    tot += e - s
ev.sort()
depth, t_prev, hist = 0, ev[0][0], defaultdict(float)
for t, dlt in ev:
    hist[min(depth, 3)] += t - t_prev
    depth += dlt
    t_prev = t
span = ev[-1][0] - ev[0][0]
print("dispatches %d span %.2f ms sum of durations %.2f ms" % (len(last), span / 1e6, tot / 1e6))
print("time with 0 / 1 / 2 / 3+ kernels in flight (ms):", ["%.2f" % (hist[k] / 1e6) for k in range(4)])
print("per queue (ms):", {k: round(v / 1e6, 2) for k, v in perq.items()})
# the kernels that run longest in total, with their average duration
agg = defaultdict(lambda: [0, 0.0])
for r in last:
    a = agg[r["Kernel_Name"][:70]]
    a[0] += 1; a[1] += int(r["End_Timestamp"]) - int(r["Start_Timestamp"])
for k, (c, t) in sorted(agg.items(), key=lambda kv: -kv[1][1])[:25]:
    print("%9.2f ms %6d x %8.1f us  %s" % (t / 1e6, c, t / c / 1e3, k))
