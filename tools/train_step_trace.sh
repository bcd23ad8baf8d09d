#!/bin/bash
# GPU box: rocprofv3 kernel trace of a training-step script; writes the ORDERED kernel list of the last step (name, duration, grid, gap to
# the previous kernel) and the per-kernel summary.
# usage: tools/train_step_trace.sh <out-prefix> <marker-kernel-substring> <script> [args...]
OUT=$1; MARK=$2; shift 2
cd /tmp && export TMPDIR=/tmp
rm -rf /tmp/trprof
rocprofv3 --kernel-trace --output-format csv -d /tmp/trprof -o t -- python "$@" > /tmp/trprof.log 2>&1
tail -2 /tmp/trprof.log
python - "$OUT" "$MARK" <<'PY'
import csv, glob, collections, sys
out, mark = sys.argv[1], sys.argv[2]
f = glob.glob("/tmp/trprof/**/t_kernel_trace.csv", recursive=True)[0]
rows = sorted(csv.DictReader(open(f)), key=lambda r: int(r["Start_Timestamp"]))
starts = [int(r["Start_Timestamp"]) for r in rows if mark in r["Kernel_Name"]]
s = starts[-1]
rows = [r for r in rows if int(r["Start_Timestamp"]) >= s]
e = max(int(r["End_Timestamp"]) for r in rows)
agg = collections.defaultdict(lambda: [0, 0])
prev = None
with open(out + "_ordered.txt", "w") as fo:
    for r in rows:
        a, b = int(r["Start_Timestamp"]), int(r["End_Timestamp"])
        k = r["Kernel_Name"].split("(")[0][:72]
        agg[k][0] += b - a; agg[k][1] += 1
        fo.write("%9.1f us  gap %7.1f  grid %7s wg %4s  %s\n" % ((b - a) / 1e3, (a - prev) / 1e3 if prev else 0.0, r.get("Grid_Size_X", r.get("Grid_Size", "?")), r.get("Workgroup_Size_X", r.get("Workgroup_Size", "?")), k))
        prev = b
tot = sum(v[0] for v in agg.values())
with open(out + "_kernels.txt", "w") as fo:
    hdr = "last step: %d launches, %.2f ms in kernels over %.2f ms wall (rocprofv3 --kernel-trace) = %.0f %%" % (len(rows), tot / 1e6, (e - s) / 1e6, 100.0 * tot / (e - s))
    print(hdr); fo.write(hdr + "\n")
    for k, v in sorted(agg.items(), key=lambda x: -x[1][0]):
        line = "%-72s n=%4d %9.3f ms %5.1f %%" % (k, v[1], v[0] / 1e6, 100 * v[0] / tot)
        fo.write(line + "\n")
    for k, v in sorted(agg.items(), key=lambda x: -x[1][0])[:30]:
        print("%-72s n=%4d %9.3f ms %5.1f %%" % (k, v[1], v[0] / 1e6, 100 * v[0] / tot))
PY
