import csv, sys, glob
from collections import defaultdict
f = glob.glob(sys.argv[1] + "/**/*counter_collection.csv", recursive=True)
if not f: print("no counter file", glob.glob(sys.argv[1] + "/**/*", recursive=True)[:20]); sys.exit()
agg = defaultdict(lambda: defaultdict(float)); cnt = defaultdict(int)
seen = set()
for r in csv.DictReader(open(f[0])):
    k = r["Kernel_Name"].replace("(anonymous namespace)::", "").replace("void ", "").replace("dm_trd32::", "").replace("dm_trd64::", "").split("(")[0][:30]
    agg[k][r["Counter_Name"]] += float(r["Counter_Value"])
    key = (k, r["Dispatch_Id"])
    if key not in seen: seen.add(key); cnt[k] += 1
names = sorted({c for v in agg.values() for c in v})
print("kernel".ljust(30), "calls", " ".join(n[-18:].rjust(18) for n in names))
for k, v in sorted(agg.items(), key=lambda x: -x[1].get(names[0], 0))[:int(sys.argv[2]) if len(sys.argv) > 2 else 12]:
    print(k.ljust(30), str(cnt[k]).rjust(5), " ".join(("%.4g" % v.get(n, 0)).rjust(18) for n in names))
