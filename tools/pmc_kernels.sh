#!/bin/bash
# GPU box: rocprofv3 PMC pass over a command; prints per-kernel averages of the requested counters (one --pmc set per pass).
# usage: tools/pmc_kernels.sh "<counters>" <out.txt> <python script> [args...]
CNT=$1; OUT=$2; shift 2
cd /tmp && export TMPDIR=/tmp
rm -rf /tmp/pmcprof
rocprofv3 --kernel-trace --pmc $CNT --output-format csv -d /tmp/pmcprof -o t -- python "$@" > /tmp/pmcprof.log 2>&1
tail -2 /tmp/pmcprof.log
python - "$OUT" <<'PY'
import csv, glob, collections, sys
f = glob.glob("/tmp/pmcprof/**/t_counter_collection.csv", recursive=True)[0]
agg = collections.defaultdict(lambda: collections.defaultdict(lambda: [0.0, 0]))
for r in csv.DictReader(open(f)):
    k = r["Kernel_Name"].split("(")[0][:60]
    agg[k][r["Counter_Name"]][0] += float(r["Counter_Value"]); agg[k][r["Counter_Name"]][1] += 1
with open(sys.argv[1], "w") as fo:
    for k, cs in sorted(agg.items(), key=lambda kv: -max(v[0] for v in kv[1].values())):
        line = "%-60s " % k + "  ".join("%s=%.4g (n=%d)" % (c, v[0] / v[1], v[1]) for c, v in sorted(cs.items()))
        fo.write(line + "\n")
for ln in open(sys.argv[1]).read().splitlines()[:12]:
    print(ln)
PY
