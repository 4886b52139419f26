#!/bin/bash
# first bench + rocprofv3 kernel-trace stats of the same command (round 1)
cd /tmp && export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
python bench.py --steps 1 --warmup 1 > gpurun_out/bench_r01.json 2> gpurun_out/bench_r01.err
tail -c 3000 gpurun_out/bench_r01.json
tail -5 gpurun_out/bench_r01.err
