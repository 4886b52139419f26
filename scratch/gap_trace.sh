#!/bin/bash
# Kernel trace of ONE timed step of the default bench (configs[1]) -> idle gaps between consecutive dispatches, attributed to
# the kernel that ran before the gap (scratch/gap_analyse.py)
cd /tmp && export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
rm -rf gpurun_out/gaptrace; mkdir -p gpurun_out/gaptrace
DRIFT_BENCH_NOPROF=1 rocprofv3 --kernel-trace --output-format csv -d gpurun_out/gaptrace -o t -- python3 bench.py --steps 1 --warmup 2 --prime-passes 0 --no-cpu-baseline --no-north-star > gpurun_out/gaptrace/stdout.json 2> gpurun_out/gaptrace/stderr.txt
f=$(find gpurun_out/gaptrace -name "*kernel_trace.csv" | head -1)
python3 scratch/gap_analyse.py "$f" > gpurun_out/gap_analysis.txt
tail -60 gpurun_out/gap_analysis.txt
rm -rf gpurun_out/gaptrace
