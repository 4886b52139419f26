#!/bin/bash
# share 0/8 of configs[2] with every product written (bitshuffle + truncate) under several settings of the writer pipeline
cd "$GRAFT_REPO_ROOT"; mkdir -p gpurun_out; export TMPDIR=/tmp
export DRIFTMI_H5_CODEC=bitshuffle
run() { tag=$1; shift; env "$@" timeout -k 10 300 python3 bench.py --workload configs2 --share 0/8 --truncate --files --outdir /dev/shm > gpurun_out/fv2_$tag.log 2>&1 || { tail -3 gpurun_out/fv2_$tag.log; rm -rf /dev/shm/tmp*; return 1; }
  tail -1 gpurun_out/fv2_$tag.log > gpurun_out/fv2_$tag.json
  python3 -c "
import json; d = json.load(open('gpurun_out/fv2_$tag.json'))
print('$tag', round(d['share_s'], 1), 's', round(d['file_bytes'] / 1e9, 1), 'GB', {k: round(v['seconds'], 1) for k, v in d['stages'].items() if isinstance(v, dict)}, 'other', round(d['stages']['other_s'], 1))"
  rm -rf /dev/shm/tmp*; sleep 8; }
run default X=1
run procs4 DRIFTMI_IO_PROCS=4
run procs8 DRIFTMI_IO_PROCS=8
run threads16 DRIFTMI_IO_THREADS=16
