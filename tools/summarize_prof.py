"""Summarises the rocprofv3 CSVs written by tools/profile.sh: per-kernel time table and per-launch
PMC averages for the crnerf kernels."""
import csv
import glob
import os
import sys
from collections import defaultdict

out = sys.argv[1]


def rows(pattern):
    for path in glob.glob(os.path.join(out, pattern), recursive=True):
        with open(path) as f:
            for r in csv.DictReader(f):
                yield r


print("== kernel stats (rocprofv3 --kernel-trace --stats)")
for r in rows("trace/**/*kernel_stats.csv"):
    print("%-70s calls %5s  total %12s ns  avg %12s ns  pct %6s" % (r.get("Name", "")[:70], r.get("Calls"), r.get("TotalDurationNs"), r.get("AverageNs"), r.get("Percentage")))

print("\n== PMC per launch (mean over dispatches)")
for sub in ("pmc_sq", "pmc_wait", "pmc_fetch", "pmc_write"):
    acc = defaultdict(lambda: defaultdict(list))
    for r in rows(sub + "/**/*counter_collection.csv"):
        acc[r["Kernel_Name"][:48]][r["Counter_Name"]].append(float(r["Counter_Value"]))
    for k, cs in acc.items():
        if "crnerf" not in k:
            continue
        for c, v in sorted(cs.items()):
            print("%-50s %-28s n=%4d mean=%16.1f" % (k, c, len(v), sum(v) / len(v)))
