#!/usr/bin/env python3
"""Condense rocprofv3 CSV output (kernel stats + PMC passes) into a small text summary for profiles/."""
import csv
import glob
import os
import sys
from collections import defaultdict

out = sys.argv[1]
KERNEL = os.environ.get("PROFILE_KERNEL", "k_render")      # dispatches whose PMC values are averaged


def find(pattern):
    return sorted(glob.glob(os.path.join(out, pattern), recursive=True))


print("== kernel stats (rocprofv3 --kernel-trace --stats) ==")
for f in find("stats/**/*kernel_stats.csv"):
    rows = list(csv.DictReader(open(f)))
    for r in rows:
        print("%-70s calls %6s  total_ns %14s  avg_ns %12s  pct %6s" % (
            r.get("Name", "")[:70], r.get("Calls"), r.get("TotalDurationNs"), r.get("AverageNs"), r.get("Percentage")))

print("\n== PMC (per-dispatch average over the %s dispatches) ==" % KERNEL)
for d in find("pmc*/"):
    for f in find(os.path.relpath(d, out) + "/**/*counter_collection.csv"):
        acc = defaultdict(lambda: [0.0, 0])
        for r in csv.DictReader(open(f)):
            name = r.get("Kernel_Name", "")
            if KERNEL not in name:
                continue
            key = (name.split("(")[0][:60], r.get("Counter_Name"))
            acc[key][0] += float(r.get("Counter_Value", 0) or 0)
            acc[key][1] += 1
        for (k, c), (s, n) in sorted(acc.items()):
            print("%-60s %-40s avg %18.1f  (n=%d)" % (k, c, s / max(n, 1), n))
