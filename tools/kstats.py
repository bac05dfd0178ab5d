#!/usr/bin/env python3
"""Summarise a rocprofv3 --kernel-trace --stats output dir: per-kernel total/avg, per step."""
import csv
import glob
import sys

d, steps = sys.argv[1], float(sys.argv[2]) if len(sys.argv) > 2 else 1.0
f = glob.glob(d + "/**/*kernel_stats.csv", recursive=True)[0]
rows = list(csv.DictReader(open(f)))
tot = sum(float(r["TotalDurationNs"]) for r in rows)
print("sum of kernel time per step: %.3f ms" % (tot / steps / 1e6))
for r in rows[:int(sys.argv[3]) if len(sys.argv) > 3 else 18]:
    print("%-70s calls/step %6.1f  ms/step %7.3f  avg %8.1f us  %5.1f%%" % (
        r["Name"][:70], float(r["Calls"]) / steps, float(r["TotalDurationNs"]) / steps / 1e6,
        float(r["AverageNs"]) / 1e3, float(r["Percentage"])))
