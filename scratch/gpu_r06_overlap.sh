#!/bin/bash
cd /tmp && export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
for st in 1 2; do
rm -rf gpurun_out/ovtrace; mkdir -p gpurun_out/ovtrace
DRIFT_BENCH_NOPROF=1 timeout -k 10 300 rocprofv3 --kernel-trace --output-format csv -d gpurun_out/ovtrace -o t -- python3 bench.py --steps 1 --warmup 2 --prime-passes 0 --streams $st --no-cpu-baseline --no-north-star > gpurun_out/ovtrace/stdout.json 2> gpurun_out/ovtrace/stderr.txt || exit 3
f=$(find gpurun_out/ovtrace -name "*kernel_trace.csv" | head -1)
echo "== streams $st"; python3 scratch/overlap_analyse.py "$f" | tee gpurun_out/r06d_overlap_s$st.txt
done
rm -rf gpurun_out/ovtrace
