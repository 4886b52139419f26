#!/bin/bash
set -o pipefail
cd "$GRAFT_REPO_ROOT"; mkdir -p gpurun_out; export TMPDIR=/tmp
run() { tag=$1; sh=$2; shift; shift; env "$@" timeout -k 10 300 python3 bench.py --workload configs2 --share $sh > gpurun_out/ap_$tag.log 2>&1 || { tail -3 gpurun_out/ap_$tag.log; return 1; }
  python3 -c "
import json;d=json.loads(open('gpurun_out/ap_$tag.log').read().strip().splitlines()[-1]);c=d['classes']
print('$tag', round(d['share_s'],1), 'kern', round(d['kernel_s'],1), 'arena_end', round(d['arena_gb_at_end'],1), 'hbm', round(d['hbm_peak_gb']), 'bt', round((c['bt_ring']['ms_per_step']+c['bt_other']['ms_per_step'])/1e3,1))"; }
run s7_ws24_b132 7/8 DRIFTMI_WORKSPACE_GB=24 DRIFT_BENCH_BEAM_GB=132 && run s0_ws24 0/8 DRIFTMI_WORKSPACE_GB=24
