#!/bin/bash
# GPU box: durations of the 13 wgrad_b_kernel launches of ONE mixed backward at 2^20 points, in launch order (rocprofv3 --kernel-trace).
cd /tmp && export TMPDIR=/tmp
rm -rf /tmp/wgb
rocprofv3 --kernel-trace --output-format csv -d /tmp/wgb -o t -- python $GRAFT_REPO_ROOT/tools/gemm_bf16_bench.py > /dev/null 2>&1
python - <<PY
import csv, glob
rows = []
for f in glob.glob("/tmp/wgb/**/t_kernel_trace.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        if "wgrad_b_kernel" in r["Kernel_Name"]:
            rows.append((int(r["Start_Timestamp"]), (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3))
rows.sort()
names = ["L1 256x93", "L2", "L3", "L4", "L5 256x93", "L5 256x256", "L6", "L7", "L8", "final", "dir 128x256", "dir 128x27", "rgb 64x128"]
last = rows[-13:]
for n, (_, d) in zip(names, last):
    print("%-14s %7.1f us" % (n, d))
print("sum %.1f us" % sum(d for _, d in last))
PY
