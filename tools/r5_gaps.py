#!/usr/bin/env python3
"""Idle time inside the last steps of a profiled bench run: union of kernel intervals vs wall, biggest gaps and what surrounds them.
usage: tools/r5_gaps.py <rocprofv3 dir> [marker-substring = stem_ring_kernel] """
import csv, glob, sys
d = sys.argv[1]
marker = sys.argv[2] if len(sys.argv) > 2 else "stem_ring_kernel"
f = glob.glob(d + "/**/*kernel_trace.csv", recursive=True)[0]
rows = sorted(((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"]) for r in csv.DictReader(open(f))))
marks = [i for i, r in enumerate(rows) if marker in r[2]]
a, b = marks[-2], marks[-1]          # one full step between two stem launches
step = rows[a:b]
t0, t1 = step[0][0], rows[b][0]
busy, cur_s, cur_e = 0, None, None
gaps = []
for s, e, n in step:
    if cur_e is None:
        cur_s, cur_e = s, e
    elif s <= cur_e:
        cur_e = max(cur_e, e)
    else:
        busy += cur_e - cur_s
        gaps.append((s - cur_e, cur_e, n))
        cur_s, cur_e = s, e
busy += cur_e - cur_s
print("step wall %.3f ms, union of kernel time %.3f ms, idle %.3f ms in %d gaps; sum of kernel durations %.3f ms (overlap %.3f ms)" % (
    (t1 - t0) / 1e6, busy / 1e6, (t1 - t0 - busy) / 1e6, len(gaps), sum(e - s for s, e, _ in step) / 1e6, (sum(e - s for s, e, _ in step) - busy) / 1e6))
hist = {}
for g, _, n in gaps:
    k = "<2us" if g < 2000 else "<5us" if g < 5000 else "<10us" if g < 10000 else "<50us" if g < 50000 else ">=50us"
    hist.setdefault(k, [0, 0])
    hist[k][0] += 1
    hist[k][1] += g
for k, (c, t) in sorted(hist.items()):
    print("  gaps %-6s: %4d, %.3f ms" % (k, c, t / 1e6))
for g, at, n in sorted(gaps, reverse=True)[:12]:
    print("  %.1f us before %s (at +%.2f ms)" % (g / 1e3, n[:70], (at - t0) / 1e6))
