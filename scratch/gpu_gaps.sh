#!/bin/bash
set -o pipefail
cd "$GRAFT_REPO_ROOT"; mkdir -p gpurun_out; export TMPDIR=/tmp
rm -rf gpurun_out/gaps
timeout -k 10 400 rocprofv3 --kernel-trace --output-format csv -d gpurun_out/gaps -- python3 bench.py --no-cpu-baseline --no-north-star --steps 1 --warmup 1 --prime-passes 0 > gpurun_out/gaps.log 2>&1 || { tail -5 gpurun_out/gaps.log; exit 1; }
python3 scratch/gaps_post.py gpurun_out/gaps | tee gpurun_out/gaps.txt
rm -rf gpurun_out/gaps
