import csv, sys
from collections import defaultdict
rows=list(csv.DictReader(open(sys.argv[1])))
ev=[(int(r['Start_Timestamp']),int(r['End_Timestamp']),r['Kernel_Name'].replace('(anonymous namespace)::','').replace('void ','').split('(')[0][:28], int(r['Grid_Size_X'])//int(r['Workgroup_Size_X'])) for r in rows]
ev.sort()
idx=[i for i,e in enumerate(ev) if e[2].startswith('bt_') ]
firsts=[idx[0]]+[idx[i] for i in range(1,len(idx)) if ev[idx[i]][0]-ev[idx[i-1]][1]>50e6]
E=ev[firsts[-1]:]
a=[i for i,e in enumerate(E) if e[2]==sys.argv[2]][0]
b=[i for i,e in enumerate(E) if e[2]==sys.argv[3]][0]
S=E[a:b]
tot=defaultdict(float); cnt=defaultdict(int)
for e in S: tot[e[2]]+=(e[1]-e[0])/1e6; cnt[e[2]]+=1
print("wall %.1f ms busy %.1f ms, %d launches"%((S[-1][1]-S[0][0])/1e6,sum(tot.values()),len(S)))
for k,v in sorted(tot.items(), key=lambda x:-x[1])[:25]: print("  %-30s %7.2f ms  x%d"%(k,v,cnt[k]))
if len(sys.argv)>4:
    t0=S[0][0]
    for e in S: print("%8.2f +%7.3f %s g=%d"%((e[0]-t0)/1e6,(e[1]-e[0])/1e6,e[2],e[3]))
