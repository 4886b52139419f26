"""Idle gaps of the device between consecutive kernels / copies of the last bench step."""
import csv, glob, sys
d = sys.argv[1]
ev = []
for f in glob.glob(d + "/**/*kernel_trace.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        ev.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"].replace("(anonymous namespace)::", "")[:70]))
for f in glob.glob(d + "/**/*memory_copy_trace.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        ev.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), "COPY " + r.get("Direction", "") + " " + r.get("Bytes", r.get("Size", ""))))
ev.sort()
# the TIMED step of `bench.py --steps 1 --warmup 1 --prime-passes 0`: the second pass over the hot path (the first is the
# warm-up; a stage pass and the BT-gen-only passes follow).  A pass starts with a burst of bt_beam* kernels.
starts = [i for i, e in enumerate(ev) if "bt_beam" in e[2]]
bursts = [starts[0]]
for a_, b_ in zip(starts[:-1], starts[1:]):
    if ev[b_][0] - ev[a_][0] > 4e6:     # a new pass: more than 4 ms after the previous beam kernel
        bursts.append(b_)
i0 = bursts[1]
i1 = bursts[2] if len(bursts) > 2 else len(ev)
seg = ev[i0:i1]
busy = 0; cur_end = seg[0][0]; gaps = []
for s, e, n in seg:
    if s > cur_end:
        gaps.append((s - cur_end, prev, n))
    if e > cur_end:
        busy += e - max(s, cur_end); cur_end = e
    prev = n
wall = seg[-1][1] - seg[0][0]
print("step wall %.2f ms, device busy %.2f ms, idle %.2f ms, %d events" % (wall / 1e6, busy / 1e6, (wall - busy) / 1e6, len(seg)))
gaps.sort(reverse=True)
print("gaps > 100 us: %d totalling %.2f ms; 10-100 us: %d totalling %.2f ms; < 10 us: %d totalling %.2f ms" % (
    sum(g[0] > 1e5 for g in gaps), sum(g[0] for g in gaps if g[0] > 1e5) / 1e6,
    sum(1e4 < g[0] <= 1e5 for g in gaps), sum(g[0] for g in gaps if 1e4 < g[0] <= 1e5) / 1e6,
    sum(g[0] <= 1e4 for g in gaps), sum(g[0] for g in gaps if g[0] <= 1e4) / 1e6))
for g in gaps[:40]:
    print("%8.1f us  after %-60s before %s" % (g[0] / 1e3, g[1], g[2]))

print("\n---- run-length timeline of the step (t ms: kernel x count, busy ms, idle-before ms)")
import re
def short(n):
    n = re.sub(r"\(.*", "", n); n = n.replace("void ", "").replace("dm_trd32::", "")
    if n in ("trd_symv_kernel", "trd_wx_kernel"): return "trd_symv/wx"   # one T1 column = both
    return n
runs = []
cur_end = seg[0][0]
for s, e, n in seg:
    idle = max(0, s - cur_end)
    nm = short(n)
    if runs and runs[-1][0] == nm and idle < 50e3:
        runs[-1][2] += 1; runs[-1][3] += e - s; runs[-1][4] += idle
    else:
        runs.append([nm, s, 1, e - s, idle])
    cur_end = max(cur_end, e)
t0 = seg[0][0]
for nm, s, cnt, busy, idle in runs:
    if busy > 3e5 or idle > 1e5:
        print("%8.2f  %-40s x%-5d busy %7.2f  idle %6.2f" % ((s - t0) / 1e6, nm[:40], cnt, busy / 1e6, idle / 1e6))

print("\n---- what follows the descriptor uploads (__amd_rocclr_copyBuffer) in the step")
import collections
h = collections.Counter()
i = 0
while i < len(seg):
    if "copyBuffer" in seg[i][2]:
        j = i
        while j < len(seg) and ("copyBuffer" in seg[j][2] or "fillBuffer" in seg[j][2]):
            j += 1
        nxt = short(seg[j][2]) if j < len(seg) else "END"
        h[nxt] += sum(1 for q in range(i, j) if "copyBuffer" in seg[q][2])
        i = j
    else:
        i += 1
for k, v in h.most_common(25):
    print("%5d  %s" % (v, k))
