#!/bin/bash
set -o pipefail
mkdir -p gpurun_out
cd "$GRAFT_REPO_ROOT"
timeout -k 10 900 python scratch/config3_share.py --config4 --out gpurun_out/r02t_config4_share.json > gpurun_out/r02t_config4_share.log 2>&1 || exit 4
tail -2 gpurun_out/r02t_config4_share.log
python - <<'PY'
import json
d=json.load(open("gpurun_out/r02t_config4_share.json"))
print(d["totals"]); print(d["kernels_ms"]); print(d["kernels_tflops"])
PY
