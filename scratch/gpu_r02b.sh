#!/bin/bash
set -o pipefail
mkdir -p gpurun_out
cd "$GRAFT_REPO_ROOT"
timeout -k 10 600 python -m pytest tests/test_gpu_btgen.py tests/test_gpu_fullsize.py tests/test_gpu_testparams.py -m gpu -x -q -s > gpurun_out/r02b_tests.log 2>&1
rc=$?
tail -5 gpurun_out/r02b_tests.log
[ $rc -ne 0 ] && exit $rc
timeout -k 10 300 python scratch/btgen_bench.py --config 2 --ranges 0:128 --out gpurun_out/r02b_bt2.json 2>&1 | tail -4 || exit 3
timeout -k 10 600 python scratch/btgen_bench.py --config 3 --ranges 200:200 100:107 --out gpurun_out/r02b_bt3.json 2>&1 | tail -8 || exit 4
