#!/bin/bash
set -o pipefail
mkdir -p gpurun_out
cd "$GRAFT_REPO_ROOT"
timeout -k 10 300 python -m pytest tests/test_gpu_primitives.py tests/test_gpu_svdkl.py -m gpu -x -q > gpurun_out/r02i_prim.log 2>&1 || { tail -20 gpurun_out/r02i_prim.log; exit 2; }
tail -2 gpurun_out/r02i_prim.log
echo "== flat tiles"; timeout -k 10 300 python scratch/gemm_bench_cov.py 2>&1 | grep TFLOP
timeout -k 10 300 python scratch/gemm_bench.py 2>&1 | grep -E "M=   92|M=   32"
echo "== no flat tiles"; DM_GEMM4_NOFLAT=1 timeout -k 10 300 python scratch/gemm_bench_cov.py 2>&1 | grep TFLOP
timeout -k 10 400 python bench.py --no-cpu-baseline > gpurun_out/r02i_bench.json 2> gpurun_out/r02i_bench.err || exit 3
python - <<'PY'
import json
d=json.loads(open("gpurun_out/r02i_bench.json").read().strip().splitlines()[-1])
print(d["value"], d["ms_per_step"], d["stage_ms"]); print(d["kernels_ms"])
PY
DRIFTMI_STORAGE=hdf5 timeout -k 10 300 python scratch/e2e_config2.py > gpurun_out/r02i_e2e_hdf5.log 2>&1 || exit 4
tail -1 gpurun_out/r02i_e2e_hdf5.log
