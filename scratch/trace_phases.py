import csv, sys
rows=list(csv.DictReader(open(sys.argv[1])))
ev=[(int(r['Start_Timestamp']),int(r['End_Timestamp']),r['Kernel_Name'].replace('(anonymous namespace)::','').replace('void ','').replace('dm_trd32::','').replace('dm_trd64::','').split('(')[0][:28], int(r['Grid_Size_X'])//int(r['Workgroup_Size_X'])) for r in rows]
ev.sort()
idx=[i for i,e in enumerate(ev) if e[2].startswith('bt_') ]
firsts=[idx[0]]+[idx[i] for i in range(1,len(idx)) if ev[idx[i]][0]-ev[idx[i-1]][1]>50e6]
start=firsts[-1]; t0=ev[start][0]
E=ev[start:]
# landmarks -> phase names
marks=[('svd_build_z_kernel','svd'),('lexmax_partial_kernel','kl:reg'),('allzero_kernel','kl:chol'),('zero_upper_kernel','kl:trsm1'),('hermitize_batched_kernel','kl:hermitize'),('trd_wx_kernel','kl:T1'),('dc_tear_kernel','kl:dc'),('zt_to_x_kernel','kl:backtr'),('jac_gather_rows_kernel','kl:sort+final')]
phase='btgen'; inkl=False
from collections import defaultdict
agg=defaultdict(lambda: defaultdict(float)); span={}
for e in E:
    for nm,ph in marks:
        if e[2]==nm:
            if ph.startswith('kl:'): inkl = inkl or ph=='kl:reg'
            if ph.startswith('kl:') and not inkl: break
            if phase!=ph and not (ph=='kl:hermitize' and phase!='kl:trsm1') and not (ph=='kl:T1' and phase not in('kl:hermitize',)) and not (ph=='kl:sort+final' and phase!='kl:backtr'):
                phase=ph
            break
    agg[phase][e[2]]+=(e[1]-e[0])/1e6
    s=span.setdefault(phase,[e[0],e[1]]); s[1]=max(s[1],e[1])
for ph in agg:
    tot=sum(agg[ph].values())
    top=sorted(agg[ph].items(), key=lambda x:-x[1])[:4]
    print("%-14s wall %6.1f busy %6.1f  "%(ph,(span[ph][1]-span[ph][0])/1e6,tot)+", ".join("%s %.1f"%(k[:22],v) for k,v in top))
