#!/usr/bin/env python3
"""Where in a profiled run do launches of a kernel fall?  usage: tools/ktrace_when.py <rocprofv3 dir> <name-substring> [marker-substring]
Counts the matching launches before the first / between consecutive launches of the marker kernel (default stem_mfma_kernel = one
per forward), i.e. initialisation vs per-step work."""
import csv
import glob
import sys

d, sub = sys.argv[1], sys.argv[2]
marker = sys.argv[3] if len(sys.argv) > 3 else "stem_mfma_kernel"
f = glob.glob(d + "/**/*kernel_trace.csv", recursive=True)[0]
rows = sorted(((int(r["Start_Timestamp"]), r["Kernel_Name"]) for r in csv.DictReader(open(f))))
marks = [t for t, n in rows if marker in n]
counts = [0] * (len(marks) + 1)
for t, n in rows:
    if sub in n:
        counts[sum(1 for m in marks if m <= t)] += 1
print("%s: %d launches; before the first %s: %d; after each of its %d launches: %s" % (sub, sum(counts), marker, counts[0], len(marks), counts[1:]))
