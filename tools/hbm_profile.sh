#!/bin/bash
# GPU box: rocprofv3 evidence for the HBM-bound stand-alone kernels (tools/hbm_kernels_bench.py): per-kernel durations from a
# --kernel-trace --stats pass, then FETCH_SIZE and WRITE_SIZE in their own --pmc passes (counters only; gpurun refuses pmc +
# tracing).  Output: gpurun_out/hbm_prof/{stats.txt,fetch.txt,write.txt}.  FETCH_SIZE on gfx950 reports half the bytes of a wide
# coalesced read (MI355X_MICROARCH.md, HBM section): the summary prints the raw KB and the doubled figure.
# Usage: hbm_profile.sh [bench script under tools/ (default hbm_kernels_bench.py)] [output name under gpurun_out/ (default hbm_prof)]
set -u
BENCH=${1:-hbm_kernels_bench.py}
OUT=$GRAFT_REPO_ROOT/gpurun_out/${2:-hbm_prof}
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
rm -rf /tmp/hbmprof
rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/hbmprof/trace -o t -- python $GRAFT_REPO_ROOT/tools/$BENCH > $OUT/bench_under_trace.log 2>&1
python - <<PY > $OUT/stats.txt
import csv, glob
for f in glob.glob("/tmp/hbmprof/trace/**/t_kernel_stats.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        if "crnerf" in r["Name"]:
            print("%-60s calls %4s avg %10.1f us  total %10.1f us" % (r["Name"].replace("void ", "").replace("crnerf::", "").split("(")[0][:60], r["Calls"], float(r["AverageNs"]) / 1e3, float(r["TotalDurationNs"]) / 1e3))
PY
cat $OUT/stats.txt
for c in FETCH_SIZE WRITE_SIZE; do
  rocprofv3 --pmc $c --output-format csv -d /tmp/hbmprof/$c -o p -- python $GRAFT_REPO_ROOT/tools/$BENCH > /dev/null 2>&1
  python - <<PY > $OUT/$c.txt
import csv, glob, collections
acc = collections.defaultdict(list)
for f in glob.glob("/tmp/hbmprof/$c/**/p_counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        if "crnerf" in r["Kernel_Name"]:
            acc[r["Kernel_Name"].replace("void ", "").replace("crnerf::", "").split("(")[0][:60]].append(float(r["Counter_Value"]))
for k, v in sorted(acc.items()):
    m = sum(v) / len(v)
    print("%-60s n=%4d  $c mean %12.1f KB per launch%s" % (k, len(v), m, "  (x2 for wide reads: %.1f KB)" % (2 * m) if "$c" == "FETCH_SIZE" else ""))
PY
  cat $OUT/$c.txt
done
