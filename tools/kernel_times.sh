#!/bin/bash
# GPU box: rocprofv3 --kernel-trace --stats of a short bench run; prints the per-kernel averages (crnerf kernels only).
# usage: tools/kernel_times.sh [bench args...]
cd /tmp && export TMPDIR=/tmp
rm -rf /tmp/ktimes
rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/ktimes -o t -- python $GRAFT_REPO_ROOT/bench.py --steps 30 --warmup 5 --no-cpu-baseline "$@" > /tmp/ktimes.log 2>&1
python - <<PY
import csv, glob
for f in glob.glob("/tmp/ktimes/**/t_kernel_stats.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        if "crnerf" in r["Name"]:
            print("%-50s calls %4s avg %9.1f us" % (r["Name"].replace("void ", "").replace("crnerf::", "").split("(")[0][:50], r["Calls"], float(r["AverageNs"]) / 1e3))
PY
