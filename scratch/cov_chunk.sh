#!/bin/bash
set -o pipefail
cd "$GRAFT_REPO_ROOT"; mkdir -p gpurun_out; export TMPDIR=/tmp
run() { tag=$1; shift; env "$@" timeout -k 10 300 python3 bench.py --workload configs2 --share 0/8 > gpurun_out/cc_$tag.log 2>&1 || { tail -3 gpurun_out/cc_$tag.log; return 1; }
  python3 -c "
import json;d=json.loads(open('gpurun_out/cc_$tag.log').read().strip().splitlines()[-1])
c=d['zgemm_cov']; print('$tag', round(d['share_s'],1), 'kern', round(d['kernel_s'],1), 'cov', c['launches'], round(c['ms']), round(c['tflops'],2), round(c['frac'],4))"; }
run x256 DM_COV_XCHUNK=256 && run x0 DM_COV_XCHUNK=0 && run x64 DM_COV_XCHUNK=64 && run x1024 DM_COV_XCHUNK=1024
