#!/bin/bash
set -o pipefail
mkdir -p gpurun_out
cd "$GRAFT_REPO_ROOT"
echo "== 4x4x4 register kernel"; timeout -k 10 300 python scratch/gemm_bench.py 2>&1 | grep TFLOP
timeout -k 10 300 python scratch/gemm_bench_cov.py 2>&1 | grep TFLOP
timeout -k 10 1000 python -m pytest tests -m gpu -x -q > gpurun_out/r02f_tests.log 2>&1
rc=$?; tail -3 gpurun_out/r02f_tests.log
[ $rc -ne 0 ] && exit $rc
timeout -k 10 400 python bench.py --no-cpu-baseline > gpurun_out/r02f_bench.json 2> gpurun_out/r02f_bench.err || exit 3
python - <<'PY'
import json
d=json.loads(open("gpurun_out/r02f_bench.json").read().strip().splitlines()[-1])
print(d["value"], d["ms_per_step"], d["stage_ms"]); print(d["kernels_ms"]); print(d["roofline"]["kernel"], d["roofline"]["frac"], d["roofline"].get("also"))
PY
