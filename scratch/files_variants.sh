#!/bin/bash
set -o pipefail
cd "$GRAFT_REPO_ROOT"; mkdir -p gpurun_out; export TMPDIR=/tmp
export DRIFTMI_H5_CODEC=bitshuffle
cat /sys/fs/cgroup/cpu.max 2>/dev/null; nproc
run() { tag=$1; shift; env "$@" timeout -k 10 400 python3 bench.py --workload configs2 --share 0/8 --truncate --files --outdir /dev/shm > gpurun_out/fv_$tag.log 2>&1 || { tail -3 gpurun_out/fv_$tag.log; return 1; }
  python3 -c "import json;d=json.loads(open('gpurun_out/fv_$tag.log').read().strip().splitlines()[-1]);print('$tag', round(d['share_s'],1), round(d['kernel_s'],1))"; }
run nice10 DRIFTMI_IO_NICE=10 && run nice0 DRIFTMI_IO_NICE=0 && run t4c3 DRIFTMI_IO_THREADS=4 DRIFTMI_IO_CHUNK_THREADS=3 && run t6c2 DRIFTMI_IO_THREADS=6 DRIFTMI_IO_CHUNK_THREADS=2
