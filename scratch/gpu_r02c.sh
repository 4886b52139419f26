#!/bin/bash
set -o pipefail
mkdir -p gpurun_out
cd "$GRAFT_REPO_ROOT"
timeout -k 10 500 python scratch/btgen_bench.py --config 3 --ranges 0:64 --skip-old --bt-gb 48 --out gpurun_out/r02c_bt3_share.json 2>&1 | tail -4 || exit 4
timeout -k 10 600 python scratch/btgen_bench.py --config 5 --ranges 300:300 300:301 --skip-old --bt-gb 48 --out gpurun_out/r02c_bt5.json 2>&1 | tail -6 || exit 5
