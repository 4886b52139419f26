#!/bin/bash
# rocprofv3 kernel statistics of the config-3 probe (BT-gen + SVD chain + KL of a few m-blocks)
cd /tmp && export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
D=gpurun_out/prof_c3
mkdir -p $D
timeout -k 10 600 rocprofv3 --kernel-trace --stats --output-format csv -d $D -o c3 -- python3 scratch/config3_probe.py "$@" --out $D/probe.json > $D/stdout.txt 2> $D/stderr.txt
tail -12 $D/stdout.txt
f=$(find $D -name "*kernel_stats.csv" | head -1)
echo "STATS FILE $f"
head -25 "$f" | cut -c1-160
find $D -name "*kernel_trace.csv" -size +20M -delete
