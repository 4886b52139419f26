#!/bin/bash
set -o pipefail
mkdir -p gpurun_out
cd "$GRAFT_REPO_ROOT"
timeout -k 10 1000 python -m pytest tests -m gpu -x -q > gpurun_out/r02j_tests.log 2>&1
rc=$?; tail -3 gpurun_out/r02j_tests.log
[ $rc -ne 0 ] && exit $rc
timeout -k 10 400 python bench.py --no-cpu-baseline > gpurun_out/r02j_bench.json 2> gpurun_out/r02j_bench.err || exit 3
python - <<'PY'
import json
d=json.loads(open("gpurun_out/r02j_bench.json").read().strip().splitlines()[-1])
print(d["value"], d["ms_per_step"], d["stage_ms"]); print(d["kernels_ms"])
PY
DRIFTMI_STORAGE=hdf5 timeout -k 10 300 python scratch/e2e_config2.py > gpurun_out/r02j_e2e_hdf5.log 2>&1 || exit 4
tail -1 gpurun_out/r02j_e2e_hdf5.log
DM_DEBUG=0 timeout -k 10 600 python scratch/config3_probe.py --m 100 --svd-batch 8 --skip-kl --out gpurun_out/r02j_probe.json 2>&1 | grep -E "SVD chain|kernel classes"
