#!/bin/bash
# MFMA-pipe counters at the NORTH-STAR workload: share 0/8 of the configs[2] job under rocprofv3 --pmc, restricted to the
# kernels of interest (--kernel-include-regex) so that a dozen / a thousand dispatches are serialised instead of 40 000.
# Two passes (one regex each); the program itself sits after `--` (no wrapper: the profiler's library initialises the GPU).
#   bash scratch/pmc_share.sh r05
tag=${1:-r05}
cd /tmp && export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
for pass in cov gram; do
  if [ $pass = cov ]; then rx='zgemm4_grouped_kernel<false, true'; else rx='jac_gram_kernel'; fi
  out=gpurun_out/pmc_share_${tag}_$pass
  rm -rf $out; mkdir -p $out
  rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE SQ_BUSY_CYCLES --kernel-include-regex "$rx" --kernel-trace \
     --output-format csv -d $out -o share -- python3 bench.py --workload configs2 --share 0/8 > $out/stdout.txt 2> $out/stderr.txt
  echo "pass $pass rc $?"; ls $out | head -5
done
python3 scratch/make_share_pmc_json.py $tag
