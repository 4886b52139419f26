#!/bin/bash
# Round-5 profile set for profiles/ (tag $1), in parts so that a gpurun call stays within its limit:
#   a  rocprofv3 kernel stats of the bench + PMC passes (FETCH_SIZE, WRITE_SIZE, MFMA busy, VALU busy: one pass each) + the
#      default bench line (N = 1: measured CPU baseline, parity object, north-star leg with two live shares)
#   b  kernel stats of a full configs[2] share (0/8) through generate() under rocprofv3 --kernel-trace --stats
#   c  MFMA-busy counters AT configs[2], dispatches restricted to the covariance kernel / jac_gram (scratch/pmc_share.sh)
#   s  all eight shares of the configs[2] job (scratch/shares_all.py) -> <tag>_configs2_shares.json
#   r  multi-rank rehearsals on one card: the whole line with 2 ranks (configs[1] sharded + north_star.job), the job alone
#      with 4 ranks, the configs[3] job (Fisher all-reduce) with 2 ranks
# Every step under its own timeout; a part stops at its first failure.
set -o pipefail
tag=${1:-r05x}
part=${2:-a}
cd "$GRAFT_REPO_ROOT"; mkdir -p gpurun_out; export TMPDIR=/tmp
if [ "$part" = "a" ]; then
TAG=$tag timeout -k 10 300 bash scratch/run_rocprof.sh > gpurun_out/prof_$tag.txt 2>&1 || exit 2
timeout -k 10 300 bash scratch/run_pmc.sh fetch_$tag FETCH_SIZE > /dev/null 2>&1 || exit 3
timeout -k 10 300 bash scratch/run_pmc.sh write_$tag WRITE_SIZE > /dev/null 2>&1 || exit 4
timeout -k 10 300 bash scratch/run_pmc.sh mfma_$tag SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES GRBM_GUI_ACTIVE > /dev/null 2>&1 || exit 5
timeout -k 10 300 bash scratch/run_pmc.sh valu_$tag SQ_ACTIVE_INST_VALU SQ_BUSY_CYCLES GRBM_GUI_ACTIVE > /dev/null 2>&1 || exit 6
python scratch/make_traffic_json.py $tag > gpurun_out/${tag}_traffic.txt 2>&1
python scratch/make_mfma_json.py $tag > gpurun_out/${tag}_mfma.txt 2>&1
python scratch/make_valu_json.py $tag > gpurun_out/${tag}_valu.txt 2>&1
cp profiles/${tag}_pmc_traffic.json profiles/${tag}_pmc_mfma.json gpurun_out/ 2>/dev/null
[ -f gpurun_out/${tag}_pmc_valu.json ] && cp gpurun_out/${tag}_pmc_valu.json profiles/
f=$(find gpurun_out/prof_$tag -name "*kernel_stats.csv" | head -1); cp "$f" gpurun_out/${tag}_bench_kernel_stats.csv
find gpurun_out/pmc_fetch_$tag gpurun_out/pmc_write_$tag gpurun_out/pmc_mfma_$tag gpurun_out/pmc_valu_$tag gpurun_out/prof_$tag -name "*.csv" -size +5M -delete
echo "$(date +%T) bench default"
DRIFT_BENCH_DETAIL=gpurun_out/${tag}_bench_detail.json timeout -k 10 1000 python bench.py 2> gpurun_out/${tag}_bench_default.err | grep "^{" > gpurun_out/${tag}_bench_default.json || exit 7   # the compact line; the full record is the detail file
fi
if [ "$part" = "b" ]; then
echo "$(date +%T) share kernel stats"
rm -rf gpurun_out/sharetrace
timeout -k 10 900 rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/sharetrace -o sh -- python3 bench.py --workload configs2 --share 0/8 > gpurun_out/${tag}_configs2_share0of8_generate.json 2> gpurun_out/share.err || exit 10
f=$(find gpurun_out/sharetrace -name "*kernel_stats.csv" | head -1); cp "$f" gpurun_out/${tag}_configs2_share_kernel_stats.csv
rm -rf gpurun_out/sharetrace
fi
if [ "$part" = "c" ]; then
echo "$(date +%T) share pmc"
timeout -k 10 1000 bash scratch/pmc_share.sh $tag > gpurun_out/${tag}_pmc_share.log 2>&1 || exit 11
cp profiles/${tag}_configs2_pmc_mfma.json gpurun_out/
rm -rf gpurun_out/pmc_share_${tag}_cov gpurun_out/pmc_share_${tag}_gram
fi
if [ "$part" = "s" ]; then
echo "$(date +%T) all shares"
timeout -k 10 1100 python scratch/shares_all.py --n 8 --out gpurun_out/${tag}_configs2_shares.json 2> gpurun_out/${tag}_shares.err || exit 12
fi
if [ "$part" = "r" ]; then
echo "$(date +%T) rehearsals"
timeout -k 10 300 python bench.py --gpus 2 --one-gpu --backend gloo --share-mmax 1 --no-cpu-baseline 2> gpurun_out/${tag}_bench2.err | grep "^{" > gpurun_out/${tag}_bench_2ranks_one_gpu_gloo_sharded_with_job.json || exit 13
timeout -k 10 300 python bench.py --workload configs2 --job --gpus 4 --one-gpu --backend gloo --share-mmax 1 2> gpurun_out/${tag}_job4.err | grep "^{" > gpurun_out/${tag}_job_configs2_4ranks_one_gpu_gloo_toy.json || exit 14
timeout -k 10 300 python bench.py --workload configs3 --job --gpus 2 --one-gpu --backend gloo --share-mmax 1 2> gpurun_out/${tag}_job2c3.err | grep "^{" > gpurun_out/${tag}_job_configs3_2ranks_one_gpu_gloo_toy.json || exit 15
fi
echo "$(date +%T) part $part done"
