#!/bin/bash
cd /tmp && export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
rm -rf gpurun_out/phtrace; mkdir -p gpurun_out/phtrace
DRIFT_BENCH_NOPROF=1 timeout -k 10 300 rocprofv3 --kernel-trace --output-format csv -d gpurun_out/phtrace -o t -- python3 bench.py --steps 1 --warmup 2 --prime-passes 0 --no-cpu-baseline --no-north-star > gpurun_out/phtrace/stdout.json 2> gpurun_out/phtrace/stderr.txt || exit 3
f=$(find gpurun_out/phtrace -name "*kernel_trace.csv" | head -1)
python3 scratch/phase_timeline.py "$f" runs > gpurun_out/r06g_phase_timeline.txt
head -30 gpurun_out/r06g_phase_timeline.txt
rm -rf gpurun_out/phtrace
