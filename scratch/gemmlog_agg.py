#!/usr/bin/env python3
"""Aggregate DM_GEMM_LOG lines (stderr of a run with DM_GEMM_LOG=1) by launch shape."""
import collections
import re
import sys

rows = []
for line in open(sys.argv[1]):
    if "GEMMLOG" not in line:
        continue
    m = re.search(r"tiles\s+(\d+) descs\s+(\d+) Mmax\s+(\d+) Nmax\s+(\d+) K\s+(\d+)\.\.\s*(\d+) rmw (\d)\s+([\d.]+) ms\s+([\d.]+) TF \(padded\s+([\d.]+)\)", line)
    if m:
        rows.append(tuple(float(x) for x in m.groups()))
tot = sum(x[7] for x in rows)
print("launches", len(rows), "total ms %.1f" % tot, "avg TF %.1f" % (sum(x[8] * x[7] for x in rows) / tot))
agg = collections.defaultdict(lambda: [0, 0.0, 0.0, 0.0])
for x in rows:
    # bucket: Mmax, Nmax rounded, K range bucket
    key = (int(x[2]) // 64 * 64, int(x[3]) // 256 * 256, int(x[5]) // 64 * 64, int(x[6]))
    a = agg[key]
    a[0] += 1; a[1] += x[7]; a[2] += x[8] * x[7]; a[3] += x[9] * x[7]
for k, a in sorted(agg.items(), key=lambda kv: -kv[1][1])[: int(sys.argv[2]) if len(sys.argv) > 2 else 25]:
    print("  Mmax~%5d Nmax~%5d Kmax~%4d rmw %d : n %4d  %8.2f ms  avg %5.1f TF (padded %5.1f)" % (k + (a[0], a[1], a[2] / a[1], a[3] / a[1])))
