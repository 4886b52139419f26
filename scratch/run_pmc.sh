#!/bin/bash
# PMC pass over the bench (separate from the kernel-trace pass, as the guide requires)
cd /tmp && export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
out=gpurun_out/pmc_$1
mkdir -p $out
shift
rocprofv3 --pmc "$@" --kernel-trace --output-format csv -d $out -o bench -- python3 bench.py --steps 1 --warmup 1 --prime-passes 0 --no-cpu-baseline --no-north-star > $out/stdout.txt 2> $out/stderr.txt
ls $out | head
