#!/bin/bash
# GPU box: rocprofv3 --kernel-trace --stats of any script; writes the per-kernel table (calls, average, total) to <out.txt> and prints its head.
# usage: tools/kstats.sh <out.txt (absolute)> <python script (absolute)> [args...]
OUT=$1; shift
cd /tmp && export TMPDIR=/tmp
rm -rf /tmp/kstats
rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/kstats -o t -- python "$@" > /tmp/kstats.log 2>&1
grep -v "rocprofv3\|amdgpu.ids" /tmp/kstats.log | tail -12
python - "$OUT" "$*" <<'PY'
import csv, glob, sys
rows = []
for f in glob.glob("/tmp/kstats/**/t_kernel_stats.csv", recursive=True):
    rows += list(csv.DictReader(open(f)))
rows.sort(key=lambda r: -float(r["TotalDurationNs"]))
with open(sys.argv[1], "w") as fo:
    fo.write("rocprofv3 --kernel-trace --stats -- python %s\n" % sys.argv[2])
    for r in rows:
        fo.write("%-90s calls %5s  avg %10.1f us  total %10.3f ms  %5s %%\n" % (r["Name"].replace("void ", "").split("(")[0][:90], r["Calls"], float(r["AverageNs"]) / 1e3,
                                                                          float(r["TotalDurationNs"]) / 1e6, r["Percentage"]))
print(open(sys.argv[1]).read()[:3000])
PY
