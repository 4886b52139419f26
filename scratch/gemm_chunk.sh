#!/bin/bash
set -o pipefail
cd "$GRAFT_REPO_ROOT"; mkdir -p gpurun_out; export TMPDIR=/tmp
run() { tag=$1; shift; env "$@" timeout -k 10 300 python3 bench.py --workload configs2 --share 0/8 > gpurun_out/gc_$tag.log 2>&1 || { tail -3 gpurun_out/gc_$tag.log; return 1; }
  python3 -c "
import json;d=json.loads(open('gpurun_out/gc_$tag.log').read().strip().splitlines()[-1])
c=d['classes']; print('$tag', round(d['share_s'],1), 'kern', round(d['kernel_s'],1), {k:(round(c[k]['ms_per_step']), round(c[k]['frac'],3)) for k in ('zgemm_grouped','gemm_grouped_realB','dgemm_grouped')})"; }
b1() { tag=$1; shift; env "$@" timeout -k 10 300 python3 bench.py --no-cpu-baseline --no-north-star > gpurun_out/gc_b_$tag.log 2>&1 || { tail -3 gpurun_out/gc_b_$tag.log; return 1; }
  python3 -c "
import json;d=json.loads(open('gpurun_out/gc_b_$tag.log').read().strip().splitlines()[-1])
c=d['roofline']['classes']; print('bench $tag', round(d['value'],1), round(d['ms_per_step'],2), round(d['roofline']['frac'],4), {k:round(c[k]['ms_per_step'],2) for k in ('zgemm_grouped','gemm_grouped_realB','dgemm_grouped')})"; }
b1 g8 DM_GEMM_GROUPM=8 && b1 g1 DM_GEMM_GROUPM=1 && b1 g4 DM_GEMM_GROUPM=4 && b1 g16 DM_GEMM_GROUPM=16 && run g8 DM_GEMM_GROUPM=8 && run g1 DM_GEMM_GROUPM=1
