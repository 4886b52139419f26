"""Summary of a DM_GEMM_LOG stderr: seconds and flops of the grouped complex GEMM launches by K range and read-modify-write."""
import re, sys
from collections import defaultdict
pat = re.compile(r"GEMMLOG tiles\s+(\d+) descs\s+(\d+) Mmax\s+(\d+) Nmax\s+(\d+) K\s+(\d+)\.\.\s*(\d+) rmw (\d)\s+([\d.]+) ms\s+([\d.]+) TF \(padded\s+([\d.]+)\)")
acc = defaultdict(lambda: [0, 0.0, 0.0, 0.0])
for line in open(sys.argv[1]):
    m = pat.search(line)
    if not m: continue
    tiles, nd, mm, nm, k0, k1, rmw, ms, tf, ptf = m.groups()
    k1 = int(k1); ms = float(ms); fl = float(tf) * ms * 1e9; pfl = float(ptf) * ms * 1e9
    kb = "K<=32" if k1 <= 32 else "K<=64" if k1 <= 64 else "K<=128" if k1 <= 128 else "K<=512" if k1 <= 512 else "K<=1024" if k1 <= 1024 else "K>1024"
    key = (kb, int(rmw), "tiles<512" if int(tiles) < 512 else "tiles<4096" if int(tiles) < 4096 else "tiles>=4096")
    a = acc[key]; a[0] += 1; a[1] += ms; a[2] += fl; a[3] += pfl
tot = sum(a[1] for a in acc.values())
print("total %.1f ms in %d launches" % (tot, sum(a[0] for a in acc.values())))
for key, a in sorted(acc.items(), key=lambda kv: -kv[1][1]):
    print("%-8s rmw %d %-12s %6d launches %9.1f ms (%4.1f%%)  %6.1f TF/s  (padded %6.1f)  avg %7.1f us" % (*key, a[0], a[1], 100 * a[1] / tot, a[2] / a[1] / 1e9, a[3] / a[1] / 1e9, 1e3 * a[1] / a[0]))
