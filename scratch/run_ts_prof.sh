#!/bin/bash
# rocprofv3 kernel stats of the bench with the two-stage tridiagonalisation on / off (development aid)
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
mkdir -p $R/gpurun_out/tsprof
for mode in ${MODES:-1}; do
  export DM_TRD_TWOSTAGE=$mode
  python $R/bench.py --steps 3 --warmup 1 --prime-passes 2 --no-cpu-baseline > $R/gpurun_out/tsprof/bench_$mode.json 2> $R/gpurun_out/tsprof/bench_$mode.err
  rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/tsprof/p$mode -o p -- python3 $R/bench.py --steps 2 --warmup 1 --prime-passes 0 --no-cpu-baseline > $R/gpurun_out/tsprof/prof_$mode.log 2>&1
  cp $(find $R/gpurun_out/tsprof/p$mode -name "*kernel_stats.csv" | head -1) $R/gpurun_out/tsprof/kernel_stats_$mode.csv
  rm -rf $R/gpurun_out/tsprof/p$mode
done
python - <<'PY'
import json,os
R=os.environ['GRAFT_REPO_ROOT']
for m in (1,):
    try:
        d=json.loads(open(f'{R}/gpurun_out/tsprof/bench_{m}.json').read().strip().splitlines()[-1])
        print('twostage',m,d['value'],d['ms_per_step'],d.get('stage_ms'))
    except Exception as e: print('fail',m,e)
PY
