#!/bin/bash
set -o pipefail
mkdir -p gpurun_out
cd "$GRAFT_REPO_ROOT"
DM_DEBUG=1 timeout -k 10 600 python scratch/config3_probe.py --m 100 --svd-batch 8 --skip-kl --out gpurun_out/r02h_probe.json > gpurun_out/r02h_probe.log 2>&1 || exit 2
grep -v GEMMLOG gpurun_out/r02h_probe.log | tail -40
DM_GEMM_LOG=1 timeout -k 10 600 python scratch/config3_probe.py --m 100 --svd-batch 8 --skip-kl --svd-once --out gpurun_out/r02h_probe2.json > gpurun_out/r02h_gemmlog.log 2>&1 || exit 3
python scratch/gemmlog_agg.py gpurun_out/r02h_gemmlog.log 30
