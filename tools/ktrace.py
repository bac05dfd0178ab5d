#!/usr/bin/env python3
"""Per-(kernel, grid) summary of a rocprofv3 --kernel-trace output dir: which launch shapes the time goes to.
usage: tools/ktrace.py <dir> <steps> [name-substring]"""
import collections
import csv
import glob
import sys

d, steps = sys.argv[1], float(sys.argv[2])
sub = sys.argv[3] if len(sys.argv) > 3 else ""
f = glob.glob(d + "/**/*kernel_trace.csv", recursive=True)[0]
acc = collections.OrderedDict()
for r in csv.DictReader(open(f)):
    name = r["Kernel_Name"]
    if sub and sub not in name:
        continue
    key = (name[:60], r.get("Grid_Size_X", r.get("Grid_Size", "")), r.get("Grid_Size_Y", ""), r.get("Grid_Size_Z", ""), r.get("Workgroup_Size_X", ""))
    dur = float(r["End_Timestamp"]) - float(r["Start_Timestamp"])
    e = acc.setdefault(key, [0, 0.0])
    e[0] += 1
    e[1] += dur
rows = sorted(acc.items(), key=lambda kv: -kv[1][1])
tot = sum(v[1] for v in acc.values())
print("total %.3f ms/step over %d launch shapes" % (tot / steps / 1e6, len(rows)))
import os
for (name, gx, gy, gz, wx), (n, t) in rows[:int(os.environ.get("KTRACE_ROWS", "60"))]:
    print("%-60s grid %9s %6s %3s wg %4s  calls/step %5.1f  avg %8.1f us  ms/step %7.3f" % (name, gx, gy, gz, wx, n / steps, t / n / 1e3, t / steps / 1e6))
