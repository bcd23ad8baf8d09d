#!/bin/bash
# GPU box: rocprofv3 kernel trace of tools/train_config4_bench.py; prints the per-kernel split of the LAST training step.
# usage: tools/train_step_profile.sh [rays=16384]
R=${1:-16384}
cd /tmp && export TMPDIR=/tmp
rm -rf /tmp/trprof
rocprofv3 --kernel-trace --output-format csv -d /tmp/trprof -o t -- python $GRAFT_REPO_ROOT/tools/train_config4_bench.py $R > /tmp/trprof.log 2>&1
tail -1 /tmp/trprof.log
python - <<PY
import csv, glob, collections
f = glob.glob("/tmp/trprof/**/t_kernel_trace.csv", recursive=True)[0]
rows = list(csv.DictReader(open(f)))
starts = [int(r["Start_Timestamp"]) for r in rows if "grid_batch" in r["Kernel_Name"]]
s, e = starts[-1], max(int(r["End_Timestamp"]) for r in rows)
agg = collections.defaultdict(lambda: [0, 0])
for r in rows:
    if int(r["Start_Timestamp"]) >= s:
        k = r["Kernel_Name"].split("(")[0][:64]
        agg[k][0] += int(r["End_Timestamp"]) - int(r["Start_Timestamp"]); agg[k][1] += 1
tot = sum(v[0] for v in agg.values())
print("last step: %.2f ms in kernels over %.2f ms wall (rocprofv3 --kernel-trace)" % (tot / 1e6, (e - s) / 1e6))
for k, v in sorted(agg.items(), key=lambda x: -x[1][0])[:24]:
    print("%-64s n=%4d %9.3f ms %5.1f %%" % (k, v[1], v[0] / 1e6, 100 * v[0] / tot))
import os
if os.environ.get("DETAIL"):      # per-launch list of one kernel in the last step, e.g. DETAIL=wgrad_kernel
    for r in rows:
        if int(r["Start_Timestamp"]) >= s and os.environ["DETAIL"] in r["Kernel_Name"]:
            print("%9.1f us  grid %s x %s x %s" % ((int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3, r.get("Grid_Size_X", r.get("Grid_Size", "?")), r.get("Grid_Size_Y", ""), r.get("Grid_Size_Z", "")))
PY
