#!/bin/bash
# configs[2] share 0/8 against the SVD / KL batch budgets (larger batches: fewer lock-step chains, two-stage tridiagonalisation in eigh_gen)
set -o pipefail
cd "$GRAFT_REPO_ROOT"; mkdir -p gpurun_out; export TMPDIR=/tmp
run() { tag=$1; shift; env "$@" timeout -k 10 300 python3 bench.py --workload configs2 --share 0/8 > gpurun_out/sb_$tag.log 2>&1 || { tail -3 gpurun_out/sb_$tag.log; return 1; }
  python3 -c "
import json;d=json.loads(open('gpurun_out/sb_$tag.log').read().strip().splitlines()[-1])
k=d['kernels_ms']
print('$tag', round(d['share_s'],1), 'kern', round(d['kernel_s'],1), 'hbm', round(d['hbm_peak_gb']), {x: round(k[x]/1e3,2) for x in ('zgemm_grouped','trd_symv','trd_wx','sb_chase','sb_q2_apply','sb_panel_qr','dc','jac_gram','jac_inner')})"; }
run s80k96w80 DRIFT_BENCH_SVD_GB=80 DRIFT_BENCH_KL_GB=96 DRIFTMI_WORKSPACE_GB=80 && run s72k96w100 DRIFT_BENCH_SVD_GB=72 DRIFT_BENCH_KL_GB=96 DRIFTMI_WORKSPACE_GB=100 && run s96k110w100 DRIFT_BENCH_SVD_GB=96 DRIFT_BENCH_KL_GB=110 DRIFTMI_WORKSPACE_GB=100
