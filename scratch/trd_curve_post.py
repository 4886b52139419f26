import csv, sys, glob
import numpy as np
n, nb = int(sys.argv[2]), int(sys.argv[3])
f = glob.glob(sys.argv[1] + "/**/*kernel_trace.csv", recursive=True)[0]
rows = list(csv.DictReader(open(f)))
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
ev = [(r["Kernel_Name"], int(r["Start_Timestamp"]), int(r["End_Timestamp"])) for r in rows]
# second repetition: from the last-but-(n-1) symv launch on
idx = [i for i, e in enumerate(ev) if "trd_symv" in e[0]]
per = n - 1
first = idx[-per]
last = idx[-1]
# walk back from `first` to the wx launch that precedes it (first column)
seg = ev[first - 1:last + 2]
k = -1
stats = {}
cur = None
for name, s, e in seg:
    short = "symv" if "trd_symv" in name else "wx" if "trd_wx" in name else "gemm" if "gemm" in name else "fill" if "fill" in name.lower() else "other"
    if short == "symv":
        k += 1
    kk = max(k, 0)
    stats.setdefault(kk, {}).setdefault(short, 0.0)
    stats[kk][short] += (e - s) / 1e3
t0, t1 = seg[0][1], seg[-1][2]
print("T1 wall ms", (t1 - t0) / 1e6)
tot = {}
for kk in stats:
    for a, b in stats[kk].items():
        tot[a] = tot.get(a, 0.0) + b
print("totals ms", {a: round(b / 1e3, 2) for a, b in tot.items()}, "sum", round(sum(tot.values()) / 1e3, 2))
ks = np.arange(per)
byts = nb * 8.0 * (n - ks - 1.0) ** 2
for lo in range(0, per, 64):
    hi = min(lo + 64, per)
    d = {a: np.mean([stats.get(q, {}).get(a, 0.0) for q in range(lo, hi)]) for a in ("symv", "wx", "gemm", "fill", "other")}
    sy = sum(stats[q]["symv"] for q in range(lo, hi))
    print("k %4d..%4d N=%4d symv %6.1f us %5.2f TB/s | wx %5.1f gemm %5.1f fill %4.1f other %4.1f (us per column)" % (
        lo, hi - 1, n - lo - 1, d["symv"], byts[lo:hi].sum() / sy / 1e6, d["wx"], d["gemm"], d["fill"], d["other"]))
