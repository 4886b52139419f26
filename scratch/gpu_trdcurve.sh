#!/bin/bash
set -o pipefail
cd "$GRAFT_REPO_ROOT"; mkdir -p gpurun_out; export TMPDIR=/tmp
N=${1:-1024}; NB=${2:-129}; TAG=${3:-a}
rm -rf gpurun_out/trdcurve_$TAG
timeout -k 10 300 rocprofv3 --kernel-trace --output-format csv -d gpurun_out/trdcurve_$TAG -- python3 scratch/trd_curve.py $N $NB > gpurun_out/trdcurve_$TAG.log 2>&1 || { tail -5 gpurun_out/trdcurve_$TAG.log; exit 1; }
python3 scratch/trd_curve_post.py gpurun_out/trdcurve_$TAG $N $NB | tee gpurun_out/trdcurve_$TAG.txt
rm -rf gpurun_out/trdcurve_$TAG
