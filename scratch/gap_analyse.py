"""Idle time between consecutive kernel dispatches of the LAST step in a rocprofv3 kernel trace (the trace holds warm-up
steps too: the last third of the dispatches is taken), by the kernel in front of the gap."""
import csv, sys
from collections import defaultdict
rows = list(csv.DictReader(open(sys.argv[1])))
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
n = len(rows)
# steps are identical sequences: find the period by the count (warmup 2 + 1 step = 3 passes, plus setup in front)
names = [r["Kernel_Name"] for r in rows]
last = rows[-(n // 3):] if len(sys.argv) < 3 else rows[-int(sys.argv[2]):]
t0, t1 = int(last[0]["Start_Timestamp"]), int(last[-1]["End_Timestamp"])
busy = 0; gaps = defaultdict(lambda: [0, 0.0]); big = []
end_prev = None; prev = None
for r in last:
    s, e = int(r["Start_Timestamp"]), int(r["End_Timestamp"])
    if end_prev is not None:
        g = s - end_prev
        if g > 0:
            key = prev[:60] + "  ->  " + r["Kernel_Name"][:60]
            gaps[key][0] += 1; gaps[key][1] += g
            if g > 100000: big.append((g, key))
        end_prev = max(end_prev, e)
    else:
        end_prev = e
    busy += e - s
    prev = r["Kernel_Name"]
span = t1 - t0
print("dispatches %d, span %.2f ms, kernel time %.2f ms, idle %.2f ms" % (len(last), span / 1e6, busy / 1e6, sum(v[1] for v in gaps.values()) / 1e6))
print("--- gaps by (kernel before -> kernel after), top 40 by total")
for k, (c, t) in sorted(gaps.items(), key=lambda kv: -kv[1][1])[:40]:
    print("%8.1f us total %5d gaps avg %7.1f us  %s" % (t / 1e3, c, t / c / 1e3, k))
print("--- single gaps over 100 us")
for g, k in sorted(big, reverse=True)[:30]:
    print("%8.1f us  %s" % (g / 1e3, k))
