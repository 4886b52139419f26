#!/bin/bash
# Shares of the configs[2] job WITH their product files (bitshuffle + LZ4, blocks truncated on the device).
#
# !! Round 6: the first version of this script ran all eight shares back to back (73 GB of files each into the box's RAM disk,
# !! plus the writer queue and its shared-memory blocks) and the GPU box was LOST five minutes in — host memory, by every sign.
# !! A lost box counts against the round (two close gpurun and the driver's GPU tiers).  So now: the shares named on the command
# !! line only (default: ONE, 0/8), a check of the host's free memory before each, and the temporary directory's removal verified
# !! after each.  usage: gpu_r06_files.sh <tag> [ranks, comma separated]
set -o pipefail
cd "$GRAFT_REPO_ROOT"; mkdir -p gpurun_out; export TMPDIR=/tmp
tag=${1:-r06z}
ranks=${2:-0}
{ nproc; df -k /tmp /dev/shm; grep MemAvailable /proc/meminfo; mount | grep -E " /tmp | /dev/shm "; } > gpurun_out/${tag}_files_env.txt 2>&1
out=/tmp
shm_avail=$(df -k --output=avail /dev/shm | tail -1)
tmp_avail=$(df -k --output=avail /tmp | tail -1)
if [ "$tmp_avail" -lt 200000000 ] && [ "$shm_avail" -gt 300000000 ]; then out=/dev/shm; fi
echo "outdir $out" >> gpurun_out/${tag}_files_env.txt
for r in ${ranks//,/ }; do
  avail_kb=$(grep MemAvailable /proc/meminfo | awk '{print $2}')
  if [ "$avail_kb" -lt 250000000 ]; then echo "only $avail_kb kB of host memory available: not starting share $r" | tee -a gpurun_out/${tag}_files_env.txt; exit 4; fi
  timeout -k 10 500 python scratch/shares_all.py --n 8 --shares $r --files --outdir $out --out gpurun_out/${tag}_configs2_share${r}of8_files.json 2>> gpurun_out/${tag}_shares_files.err || exit 5
  left=$(ls -d $out/tmp* 2>/dev/null | wc -l)
  echo "share $r done; temporary directories left under $out: $left; $(grep MemAvailable /proc/meminfo)" | tee -a gpurun_out/${tag}_files_env.txt
  if [ "$left" -gt 0 ]; then echo "temporary product directories were not removed: stopping"; exit 6; fi
  sleep 10
done
tail -3 gpurun_out/${tag}_shares_files.err
