"""Idle-gap report of one bench step from a rocprofv3 kernel trace reduced to start,end,name lines (development aid)."""
import re, sys
from collections import defaultdict
rows=[l.rstrip("\n").split(",",2) for l in open(sys.argv[1] if len(sys.argv)>1 else 'gpurun_out/gap_trace.csv')]
rows=[(int(a),int(b),c) for a,b,c in rows]
def short(k):
    k=re.sub(r"dm_trd\d+::","",k.replace("(anonymous namespace)::","").replace("void ",""))
    return k.split("(")[0][:34]
half=len(rows)//2
# second step starts at the first bt_beam_kernel at/after half-5
st=next(i for i in range(half-20,len(rows)) if 'bt_beam' in rows[i][2] and i>=half-5)
R=rows[st-1:]
t0=R[0][0]; cur=R[0][0]; busy=0; gaps=[]
gapby=defaultdict(lambda:[0,0.0])
for i,(s,e,k) in enumerate(R):
    if s>cur:
        gaps.append((i,(s-cur)/1e3,(s-t0)/1e6)); gapby[short(k)][0]+=1; gapby[short(k)][1]+=(s-cur)/1e3
    busy+=max(0,e-max(s,cur)); cur=max(cur,e)
span=(R[-1][1]-t0)/1e6
print("span %.2f busy %.2f idle %.2f ms"%(span,busy/1e6,span-busy/1e6))
for k,v in sorted(gapby.items(), key=lambda kv:-kv[1][1])[:12]: print("%-36s n %4d %8.1f us avg %.1f"%(k,v[0],v[1],v[1]/v[0]))
for i,g,t in gaps:
    if g>60: print("t=%7.2f gap %7.1f | %s -> %s, %s"%(t,g,short(R[i-1][2]),short(R[i][2]),short(R[i+1][2]) if i+1<len(R) else ""))
