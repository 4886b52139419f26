#!/bin/bash
# Profile set of round 2 for profiles/: kernel-trace stats; FETCH_SIZE, WRITE_SIZE and the MFMA-busy counters
# each in a pass of their own (--pmc never together with sys/hip traces: gpurun refuses that combination).
#   bash scratch/run_profiles_r02.sh <tag>
tag=$1
TAG=$tag bash scratch/run_rocprof.sh > gpurun_out/prof_$tag.txt 2>&1
bash scratch/run_pmc.sh fetch_$tag FETCH_SIZE > /dev/null 2>&1
bash scratch/run_pmc.sh write_$tag WRITE_SIZE > /dev/null 2>&1
bash scratch/run_pmc.sh mfma_$tag SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES GRBM_GUI_ACTIVE > /dev/null 2>&1
python scratch/pmc_summary.py gpurun_out/pmc_fetch_$tag 6
python scratch/pmc_summary.py gpurun_out/pmc_write_$tag 6
python scratch/pmc_summary.py gpurun_out/pmc_mfma_$tag 10
