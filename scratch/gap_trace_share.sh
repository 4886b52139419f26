#!/bin/bash
# Kernel trace of a configs[2] share (default 0/8) -> idle gaps by (kernel before -> kernel after) (scratch/gap_analyse.py)
cd /tmp && export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
rm -rf gpurun_out/gaptrace; mkdir -p gpurun_out/gaptrace
DRIFT_BENCH_NOPROF=1 rocprofv3 --kernel-trace --output-format csv -d gpurun_out/gaptrace -o t -- python3 bench.py --workload configs2 --share ${1:-0/8} > gpurun_out/gaptrace/stdout.json 2> gpurun_out/gaptrace/stderr.txt
f=$(find gpurun_out/gaptrace -name "*kernel_trace.csv" | head -1)
n=$(wc -l < "$f")
python3 scratch/gap_analyse.py "$f" $((n - 1)) > gpurun_out/gap_analysis_share.txt
head -50 gpurun_out/gap_analysis_share.txt
rm -rf gpurun_out/gaptrace
