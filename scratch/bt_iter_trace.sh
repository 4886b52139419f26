#!/bin/bash
# per-dispatch durations of the real-B grouped GEMM launches of one BT-gen rank call; $1 = m range, $2 = iter
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
rm -rf gpurun_out/bttrace
timeout -k 10 600 rocprofv3 --kernel-trace --output-format csv -d gpurun_out/bttrace -o bt -- python3 scratch/bt_iter_bench.py --config 3 --ranges $1 --iters $2 --reps 1 > gpurun_out/bttrace.log 2>&1
f=$(find gpurun_out/bttrace -name "*kernel_trace.csv" | head -1)
python3 - "$f" <<'PY'
import csv, sys
rows = list(csv.DictReader(open(sys.argv[1])))
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
half = len(rows) // 2
n = 0
for r in rows[half:]:
    nm = r["Kernel_Name"]
    if "zgemm4_grouped_kernel<true" in nm or "bt_refine_update" in nm or "bt_fused" in nm or "dgemm" in nm:
        dur = (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e6
        short = nm.replace("(anonymous namespace)::", "").replace("void ", "")[:40]
        print("%-40s grid %8s  %9.2f ms" % (short, r.get("Grid_Size_X", r.get("Grid_Size", "?")), dur))
        n += 1
        if n > 110: break
PY
rm -rf gpurun_out/bttrace
