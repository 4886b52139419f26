#!/bin/bash
set -o pipefail
cd "$GRAFT_REPO_ROOT"; mkdir -p gpurun_out; export TMPDIR=/tmp
rm -rf gpurun_out/stats
timeout -k 10 400 rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/stats -o b -- python3 bench.py --no-cpu-baseline --steps 1 --warmup 1 --prime-passes 0 > gpurun_out/stats.log 2>&1 || { tail -5 gpurun_out/stats.log; exit 1; }
python3 - "$@" <<'PY'
import csv,re,glob,sys
f=glob.glob('gpurun_out/stats/**/*kernel_stats.csv', recursive=True)[0]
rows=list(csv.DictReader(open(f)))
tot=sum(float(r['TotalDurationNs']) for r in rows)
print("total kernel ms per step %.1f"%(tot/2e6))
pat=sys.argv[1] if len(sys.argv)>1 else None
for r in rows[:int(sys.argv[2]) if len(sys.argv)>2 else 40]:
    n=re.sub(r'\(\(.*|\(dm_.*|\(HIP.*|\(double.*','',r['Name']).replace('void ','').replace('(anonymous namespace)::','').replace('dm_trd32::','')
    if pat and not re.search(pat, n): continue
    print("%-52s calls/step %6.0f  %7.2f ms  avg %8.1f us"%(n[:52], int(r['Calls'])/2, float(r['TotalDurationNs'])/2e6, float(r['AverageNs'])/1e3))
PY
rm -rf gpurun_out/stats
