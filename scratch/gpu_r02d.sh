#!/bin/bash
set -o pipefail
mkdir -p gpurun_out
cd "$GRAFT_REPO_ROOT"
DRIFTMI_STORAGE=hdf5 timeout -k 10 300 python scratch/e2e_config2.py > gpurun_out/r02d_e2e_hdf5.log 2>&1 || exit 2
tail -2 gpurun_out/r02d_e2e_hdf5.log
DRIFTMI_STORAGE=npz timeout -k 10 300 python scratch/e2e_config2.py > gpurun_out/r02d_e2e_npz.log 2>&1 || exit 3
tail -1 gpurun_out/r02d_e2e_npz.log
timeout -k 10 900 python scratch/config3_share.py --config4 --out gpurun_out/r02d_config4_share.json > gpurun_out/r02d_config4_share.log 2>&1 || exit 4
tail -3 gpurun_out/r02d_config4_share.log
