#!/bin/bash
set -o pipefail
cd "$GRAFT_REPO_ROOT"; mkdir -p gpurun_out; export TMPDIR=/tmp
for v in "$@"; do
rm -rf gpurun_out/btprof
export DM_FDFT_SHARED=$v
timeout -k 10 500 rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/btprof -o bt -- python3 scratch/btgen_bench.py --config 3 --ranges 0:64 --skip-old --out gpurun_out/btprof.json > gpurun_out/btprof.log 2>&1 || { tail -5 gpurun_out/btprof.log; exit 1; }
echo "DM_FDFT_SHARED=$v"; grep fused gpurun_out/btprof.log
python3 - <<'PY'
import csv,re,glob
f=glob.glob('gpurun_out/btprof/**/*kernel_stats.csv', recursive=True)[0]
for r in list(csv.DictReader(open(f)))[:4]:
    n=re.sub(r'\(\(.*','',r['Name']).replace('void ','').replace('(anonymous namespace)::','')
    print("  %-50s calls %6s total %9.1f ms"%(n[:50], r['Calls'], float(r['TotalDurationNs'])/1e6))
PY
done
rm -rf gpurun_out/btprof
