#!/bin/bash
# all eight shares of the configs[2] job WITH their product files (bitshuffle + LZ4, blocks truncated on the device)
set -o pipefail
cd "$GRAFT_REPO_ROOT"; mkdir -p gpurun_out; export TMPDIR=/tmp
tag=${1:-r06z}
{ nproc; df -k /tmp /dev/shm; grep MemAvailable /proc/meminfo; mount | grep -E " /tmp | /dev/shm " ; } > gpurun_out/${tag}_files_env.txt 2>&1
out=/tmp
shm_avail=$(df -k --output=avail /dev/shm | tail -1)
tmp_avail=$(df -k --output=avail /tmp | tail -1)
if [ "$tmp_avail" -lt 200000000 ] && [ "$shm_avail" -gt 300000000 ]; then out=/dev/shm; fi
echo "outdir $out" >> gpurun_out/${tag}_files_env.txt
timeout -k 10 1150 python scratch/shares_all.py --n 8 --files --outdir $out --out gpurun_out/${tag}_configs2_shares_files.json 2> gpurun_out/${tag}_shares_files.err
tail -3 gpurun_out/${tag}_shares_files.err
cat gpurun_out/${tag}_files_env.txt
