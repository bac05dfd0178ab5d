#!/usr/bin/env python3
"""Wrap tools/traffic.sh outputs into the profiles/ format bench.py cites:
    tools/traffic_wrap.py <out.json> <workload text> <clips_per_launch>=<in.json> [<clips_per_launch>=<in.json> ...]
(run in the build container: stamps the commit the kernels were measured at)."""
import json
import subprocess
import sys

dst, workload = sys.argv[1], sys.argv[2]
head = subprocess.check_output(["git", "rev-parse", "--short", "HEAD"]).decode().strip()
out = {}
for item in sys.argv[3:]:
    clips, src = item.split("=", 1)
    rec = json.load(open(src))
    rec["workload"] = workload
    rec["clips_per_launch"] = int(clips)
    out[clips] = rec
json.dump({"by_clips_per_launch": out, "measured_at_commit": head}, open(dst, "w"), indent=1)
print(dst, head)
