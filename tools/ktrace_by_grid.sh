#!/bin/bash
# GPU box: rocprofv3 --kernel-trace of any script; per (kernel, grid) average duration in launch order -- tells the jobs of a multi-launch
# operator (the 13 weight-gradient blocks of one backward) apart.  usage: tools/ktrace_by_grid.sh <out.txt (absolute)> <filter> <script (absolute)> [args...]
OUT=$1; shift; FILTER=$1; shift
cd /tmp && export TMPDIR=/tmp
rm -rf /tmp/ktrace
rocprofv3 --kernel-trace --output-format csv -d /tmp/ktrace -o t -- python "$@" > /tmp/ktrace.log 2>&1
grep -v "rocprofv3\|amdgpu.ids" /tmp/ktrace.log | tail -12
python - "$OUT" "$FILTER" "$*" <<'PY'
import csv, glob, sys
from collections import OrderedDict
rows = []
for f in glob.glob("/tmp/ktrace/**/t_kernel_trace.csv", recursive=True):
    rows += list(csv.DictReader(open(f)))
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
agg = OrderedDict()
for r in rows:
    if sys.argv[2] not in r["Kernel_Name"]:
        continue
    key = (r["Kernel_Name"].split("(")[0][-60:], r["Grid_Size_X"], r["Grid_Size_Y"], r["Grid_Size_Z"])
    agg.setdefault(key, []).append((int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3)
with open(sys.argv[1], "w") as fo:
    fo.write("rocprofv3 --kernel-trace -- python %s   (filter %s; grid in work-items)\n" % (sys.argv[3], sys.argv[2]))
    for k, v in agg.items():
        fo.write("%-60s grid %8s x %4s x %4s  calls %4d  avg %9.1f us  min %9.1f us\n" % (k[0], k[1], k[2], k[3], len(v), sum(v) / len(v), min(v)))
    import os
    n = int(os.environ.get("KTRACE_SEQ", "0"))       # KTRACE_SEQ=N: also the first and the last N matching launches, in launch order
    if n:
        sel = [r for r in rows if sys.argv[2] in r["Kernel_Name"]]
        for tag, part in (("first", sel[:n]), ("last", sel[-n:])):
            fo.write("%s %d launches: %s\n" % (tag, n, "  ".join("%s %.0f" % (r["Kernel_Name"].split("(")[0].split("::")[-1][:14], (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3) for r in part)))
print(open(sys.argv[1]).read()[:6000])
PY
