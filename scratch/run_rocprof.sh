#!/bin/bash
cd /tmp && export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/prof_${TAG:-r01x}
rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/prof_${TAG:-r01x} -o bench -- python3 bench.py --steps 1 --warmup 1 --prime-passes 0 --no-cpu-baseline --no-north-star > gpurun_out/prof_${TAG:-r01x}/bench_stdout.json 2> gpurun_out/prof_${TAG:-r01x}/stderr.txt
ls -R gpurun_out/prof_${TAG:-r01x} | head -30
f=$(find gpurun_out/prof_${TAG:-r01x} -name "*kernel_stats.csv" | head -1)
echo "STATS FILE $f"
head -12 "$f" | cut -c1-150
# keep only the small summaries (the raw trace is large)
find gpurun_out/prof_${TAG:-r01x} -name "*kernel_trace.csv" -size +20M -delete
