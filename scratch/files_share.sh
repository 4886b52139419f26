#!/bin/bash
# file output of a configs[2] share: 25 m-blocks with the product files written (tmpfs/scratch of the box) vs discarded
set -o pipefail
cd "$GRAFT_REPO_ROOT"; mkdir -p gpurun_out; export TMPDIR=/tmp
tag=${1:-files}
nproc > gpurun_out/${tag}_env.txt; df -h /tmp >> gpurun_out/${tag}_env.txt; grep MemAvailable /proc/meminfo >> gpurun_out/${tag}_env.txt
timeout -k 10 500 python3 bench.py --workload configs2 --share 0/40 --files > gpurun_out/${tag}_default.log 2>&1 || { tail -5 gpurun_out/${tag}_default.log; exit 1; }
tail -1 gpurun_out/${tag}_default.log | cut -c1-1500
DRIFTMI_IO_CHUNK_THREADS=1 timeout -k 10 500 python3 bench.py --workload configs2 --share 0/40 --files > gpurun_out/${tag}_ct1.log 2>&1 || { tail -5 gpurun_out/${tag}_ct1.log; exit 1; }
tail -1 gpurun_out/${tag}_ct1.log | cut -c1-600
DRIFTMI_IO_CHUNK_THREADS=8 DRIFTMI_IO_THREADS=4 timeout -k 10 500 python3 bench.py --workload configs2 --share 0/40 --files > gpurun_out/${tag}_ct8.log 2>&1 || { tail -5 gpurun_out/${tag}_ct8.log; exit 1; }
tail -1 gpurun_out/${tag}_ct8.log | cut -c1-600
