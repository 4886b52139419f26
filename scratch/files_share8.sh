#!/bin/bash
# file output of a full configs[2] share 0/8, bitshuffle + truncate (the reference's production setting): files vs no files
set -o pipefail
cd "$GRAFT_REPO_ROOT"; mkdir -p gpurun_out; export TMPDIR=/tmp
tag=${1:-files8}
share=${2:-0/8}
{ nproc; df -k /tmp /dev/shm; grep MemAvailable /proc/meminfo; } > gpurun_out/${tag}_env.txt
out=/tmp
shm_avail=$(df -k --output=avail /dev/shm | tail -1)
tmp_avail=$(df -k --output=avail /tmp | tail -1)
if [ "$tmp_avail" -lt 200000000 ] && [ "$shm_avail" -gt 300000000 ]; then out=/dev/shm; fi
echo "outdir $out" >> gpurun_out/${tag}_env.txt
mode=${3:-bshuf}
if [ "$mode" = bshuf ]; then export DRIFTMI_H5_CODEC=bitshuffle; trunc=--truncate; else export DRIFTMI_H5_CODEC=lzf; trunc=; fi
timeout -k 10 400 python3 bench.py --workload configs2 --share $share $trunc > gpurun_out/${tag}_nofiles.log 2>&1 || { tail -5 gpurun_out/${tag}_nofiles.log; exit 1; }
tail -1 gpurun_out/${tag}_nofiles.log > gpurun_out/${tag}_nofiles.json
echo "nofiles done" 
timeout -k 10 600 python3 bench.py --workload configs2 --share $share $trunc --files --outdir $out > gpurun_out/${tag}_files.log 2>&1 || { tail -5 gpurun_out/${tag}_files.log; rm -rf $out/tmp*; exit 1; }
tail -1 gpurun_out/${tag}_files.log > gpurun_out/${tag}_files.json
python3 - <<PY
import json
a = json.load(open("gpurun_out/${tag}_nofiles.json")); b = json.load(open("gpurun_out/${tag}_files.json"))
print("nofiles %.1f s, files %.1f s, %.1f GB, %.2f GB/s over the difference, %.2f GB/s over the whole share"
      % (a["share_s"], b["share_s"], b["file_bytes"] / 1e9, b["file_bytes"] / 1e9 / max(b["share_s"] - a["share_s"], 1e-9), b["file_bytes"] / 1e9 / b["share_s"]))
PY
