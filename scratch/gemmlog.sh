#!/bin/bash
set -o pipefail
cd "$GRAFT_REPO_ROOT"; mkdir -p gpurun_out; export TMPDIR=/tmp
DM_GEMM_LOG=1 timeout -k 10 300 python3 bench.py --no-cpu-baseline --no-north-star --steps 1 --warmup 1 --prime-passes 0 > gpurun_out/gemmlog.out 2> gpurun_out/gemmlog.err || { tail -3 gpurun_out/gemmlog.err; exit 1; }
grep -c GEMMLOG gpurun_out/gemmlog.err
