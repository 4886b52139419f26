#!/bin/bash
set -o pipefail
mkdir -p gpurun_out
cd "$GRAFT_REPO_ROOT"
for rot in 1 0; do
  DM_GEMM4=1 DM_GEMM4_ROT=$rot timeout -k 10 300 python -m pytest tests/test_gpu_primitives.py -m gpu -x -q -k "gemm or zgemm" > gpurun_out/r02e_gemm4_rot$rot.log 2>&1
  echo "rot=$rot rc=$?"; tail -3 gpurun_out/r02e_gemm4_rot$rot.log
done
echo "== 16x16x4 LDS kernel"; DM_GEMM4=0 timeout -k 10 300 python scratch/gemm_bench.py 2>&1 | grep TFLOP
echo "== 4x4x4 register kernel"; DM_GEMM4=1 timeout -k 10 300 python scratch/gemm_bench.py 2>&1 | grep TFLOP
