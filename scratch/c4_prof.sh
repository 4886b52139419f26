#!/bin/bash
# one real configs[4] block through the product classes under rocprofv3 --kernel-trace --stats; $1 = tag
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
rm -rf gpurun_out/c4prof
timeout -k 10 1000 rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/c4prof -o c4 -- python3 bench.py --workload configs4 --m 300 > gpurun_out/$1_configs4_block_m300.json 2> gpurun_out/$1_configs4_block_m300.log
f=$(find gpurun_out/c4prof -name "*kernel_stats.csv" | head -1)
cp "$f" gpurun_out/$1_configs4_block_m300_kernel_stats.csv
rm -rf gpurun_out/c4prof
tail -12 gpurun_out/$1_configs4_block_m300.log
head -12 gpurun_out/$1_configs4_block_m300_kernel_stats.csv | cut -c1-150
