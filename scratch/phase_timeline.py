"""Wall-time phases of ONE configs[1] step from a rocprofv3 kernel trace: the dispatches of the last third in time order,
cut into runs by a kernel-name -> phase map; per phase: wall span (first start to last end), kernel time, launches."""
import csv
import sys
from collections import OrderedDict

PH = [("bt_", "btgen"), ("svd_", "svd"), ("jac_", "svd-jacobi"), ("potf2", "chol"), ("panel_trsm", "chol"), ("diag_solve", "trsm"),
      ("sb_panel", "stage1-qr"), ("larft", "stage1-side"), ("sb_sum", "stage1-main"), ("sb_s_kernel", "stage1-main"),
      ("sb_band", "chase"), ("sb_chase", "chase"), ("sb_vd", "stage1-pre"), ("sb_diag", "stage1-pre"), ("sb_q2", "q2"),
      ("dc_", "dc"), ("ql_kernel", "dc"), ("trd_small", "trd_small"), ("trd_", "trd1"), ("cov_", "cov"), ("zgemm4_grouped_kernel<false, false, 2", "cov")]


def phase_of(name, prev):
    for k, p in PH:
        if k in name:
            return p
    return prev   # grouped GEMMs, copies, fills: part of whatever phase they sit in


rows = list(csv.DictReader(open(sys.argv[1])))
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
last = rows[-(len(rows) // 3):]
runs = []
cur = None
for r in last:
    s, e = int(r["Start_Timestamp"]), int(r["End_Timestamp"])
    p = phase_of(r["Kernel_Name"], cur[0] if cur else "start")
    if cur is None or p != cur[0]:
        cur = [p, s, e, 0.0, 0, 0, 0.0]
        runs.append(cur)
    cur[2] = max(cur[2], e)
    cur[3] += e - s
    cur[4] += 1
    if "copyBuffer" in r["Kernel_Name"] or "fillBuffer" in r["Kernel_Name"]:
        cur[5] += 1
        cur[6] += e - s
t0 = runs[0][1]
agg = OrderedDict()
print("span of the step: %.2f ms" % ((runs[-1][2] - t0) / 1e6))
for i, (p, s, e, k, n, nc, tc) in enumerate(runs):
    nxt = runs[i + 1][1] if i + 1 < len(runs) else e
    a = agg.setdefault(p, [0.0, 0.0, 0, 0, 0.0])
    a[0] += (nxt - s) / 1e6; a[1] += k / 1e6; a[2] += n; a[3] += nc; a[4] += tc / 1e6
for p, (w, k, n, nc, tc) in sorted(agg.items(), key=lambda kv: -kv[1][0]):
    print("%-14s wall %7.2f ms  kernels %7.2f ms  launches %5d  of them copies / fills %4d (%.2f ms)" % (p, w, k, n, nc, tc))
if len(sys.argv) > 2:
    for p, s, e, k, n, nc, tc in runs:
        print("%9.3f ms  %-14s wall %7.3f kernels %7.3f launches %4d" % ((s - t0) / 1e6, p, (e - s) / 1e6, k / 1e6, n))
