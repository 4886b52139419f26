#!/bin/bash
# round-2 GPU pass A: the whole GPU suite (with the config-3/4/5 tests), the default bench line and the
# two-rank rehearsal on one card
set -o pipefail
mkdir -p gpurun_out
cd "$GRAFT_REPO_ROOT"
timeout -k 10 1000 python -m pytest tests -m gpu -x -q -s > gpurun_out/r02a_tests.log 2>&1
rc=$?
tail -5 gpurun_out/r02a_tests.log
[ $rc -ne 0 ] && exit $rc
timeout -k 10 400 python bench.py > gpurun_out/r02a_bench.json 2> gpurun_out/r02a_bench.err || exit 3
tail -c 600 gpurun_out/r02a_bench.json
timeout -k 10 300 python bench.py --gpus 2 --one-gpu --backend gloo --no-cpu-baseline > gpurun_out/r02a_bench2.json 2> gpurun_out/r02a_bench2.err || exit 4
timeout -k 10 300 python bench.py --gpus 2 --one-gpu --backend gloo --mode sharded --no-cpu-baseline > gpurun_out/r02a_bench2s.json 2> gpurun_out/r02a_bench2s.err || exit 5
python - <<'PY'
import json
for f in ("r02a_bench.json","r02a_bench2.json","r02a_bench2s.json"):
    d=json.loads(open("gpurun_out/"+f).read().strip().splitlines()[-1])
    print(f, d["value"], d["n_gpus"], d["ms_per_step"], d["scaling"], d["stage_ms"])
PY
