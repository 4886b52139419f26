#!/bin/bash
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
N=${1:-8192}
mkdir -p $R/gpurun_out/bigprof
rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/bigprof/p -o p -- python3 $R/scratch/twostage_big.py $N 1 ${2:-1} > $R/gpurun_out/bigprof/log.txt 2>&1
cp $(find $R/gpurun_out/bigprof/p -name "*kernel_stats.csv" | head -1) $R/gpurun_out/bigprof/kernel_stats.csv
rm -rf $R/gpurun_out/bigprof/p
tail -3 $R/gpurun_out/bigprof/log.txt
