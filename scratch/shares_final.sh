#!/bin/bash
set -o pipefail
cd "$GRAFT_REPO_ROOT"; mkdir -p gpurun_out; export TMPDIR=/tmp
tag=${1:-r04q}
for r in ${SHARES:-1 2 4 7}; do
  timeout -k 10 300 python3 bench.py --workload configs2 --share $r/8 > gpurun_out/${tag}_configs2_share${r}of8_generate.json 2> gpurun_out/share$r.err || { tail -3 gpurun_out/share$r.err; exit 1; }
  python3 -c "
import json;d=json.loads(open('gpurun_out/${tag}_configs2_share${r}of8_generate.json').read().strip().splitlines()[-1])
c=d['classes']; print($r, d['config']['workload'].split('generate(): ')[1][:40], round(d['share_s'],1), 'kern', round(d['kernel_s'],1), 'cov', round(d['zgemm_cov']['frac'],3), 'zgemm', round(c['zgemm_grouped']['frac'],3), 'bt', round((c['bt_ring']['ms_per_step']+c['bt_other']['ms_per_step']+c['gemm_grouped_realB']['ms_per_step'])/1e3,1), 'hbm', round(d['hbm_peak_gb']))"
done
