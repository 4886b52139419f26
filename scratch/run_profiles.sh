#!/bin/bash
# Full profile set for profiles/: kernel-trace stats, then FETCH_SIZE and WRITE_SIZE in their own passes.
tag=$1
TAG=$tag bash scratch/run_rocprof.sh > /dev/null 2>&1
bash scratch/run_pmc.sh fetch_$tag FETCH_SIZE > /dev/null 2>&1
bash scratch/run_pmc.sh write_$tag WRITE_SIZE > /dev/null 2>&1
python scratch/pmc_summary.py gpurun_out/pmc_fetch_$tag 6
python scratch/pmc_summary.py gpurun_out/pmc_write_$tag 6
python bench.py 2>&1 | tail -1 > gpurun_out/bench_default_$tag.json
cut -c1-300 gpurun_out/bench_default_$tag.json
