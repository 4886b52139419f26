import csv, sys
from collections import defaultdict
rows=list(csv.DictReader(open(sys.argv[1])))
ev=[(int(r['Start_Timestamp']),int(r['End_Timestamp']),r['Kernel_Name'].replace('(anonymous namespace)::','').replace('void ','').replace('dm_trd32::','').replace('dm_trd64::','').split('(')[0][:28], int(r['Grid_Size_X'])//int(r['Workgroup_Size_X']), int(r['Grid_Size_Y'])) for r in rows]
ev.sort()
idx=[i for i,e in enumerate(ev) if e[2].startswith('bt_') ]
firsts=[idx[0]]+[idx[i] for i in range(1,len(idx)) if ev[idx[i]][0]-ev[idx[i-1]][1]>50e6]
start=firsts[-1]; t0=ev[start][0]
E=ev[start:]
names=sys.argv[2].split(',')
for nm in names:
    L=[e for e in E if e[2]==nm]
    d=defaultdict(list)
    for e in L: d[e[4]].append((e[1]-e[0])/1e3)
    for ny,v in d.items(): print(nm,"nprob",ny,"calls",len(v),"total ms %.1f"%(sum(v)/1e3),"max us %.0f"%max(v), "first8", [round(x) for x in v[:8]])
    big=max(d.keys(), key=lambda k: sum(d[k])) if d else None
    if big is not None and len(d[big])>300:
        h=d[big]
        print("   per-100 avg us:", [round(sum(h[a:a+100])/len(h[a:a+100])) for a in range(0,len(h),100)])
tot=defaultdict(float)
for e in E: tot[e[2]]+=(e[1]-e[0])/1e6
print("span %.1f ms busy %.1f ms"%((E[-1][1]-t0)/1e6, sum(tot.values())))
for k,v in sorted(tot.items(), key=lambda x:-x[1])[:14]: print("  %-30s %.1f"%(k,v))
