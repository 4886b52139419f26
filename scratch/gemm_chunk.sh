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
b1 c32 DM_GEMM_XCHUNK=32 DM_GEMMR_XCHUNK=64 DM_DGEMM_XCHUNK=64 && b1 c16r16d16 DM_GEMM_XCHUNK=16 DM_GEMMR_XCHUNK=16 DM_DGEMM_XCHUNK=16 && b1 c256r32d32 DM_GEMM_XCHUNK=256 DM_GEMMR_XCHUNK=32 DM_DGEMM_XCHUNK=32 && b1 c4r4d4 DM_GEMM_XCHUNK=4 DM_GEMMR_XCHUNK=4 DM_DGEMM_XCHUNK=4 && run c32r32 DM_GEMM_XCHUNK=32 DM_GEMMR_XCHUNK=32 DM_DGEMM_XCHUNK=32 DM_COV_XCHUNK=32
