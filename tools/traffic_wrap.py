#!/usr/bin/env python3
"""Wrap tools/traffic.sh outputs into the profiles/ format bench.py cites: usage tools/traffic_wrap.py <in.json> <out.json> <clips_per_launch> <workload text>
(run in the build container: stamps the commit the kernels were measured at)."""
import json
import subprocess
import sys

src, dst, clips, workload = sys.argv[1], sys.argv[2], sys.argv[3], sys.argv[4]
rec = json.load(open(src))
rec["workload"] = workload
rec["clips_per_launch"] = int(clips)
head = subprocess.check_output(["git", "rev-parse", "--short", "HEAD"]).decode().strip()
json.dump({"by_clips_per_launch": {clips: rec}, "measured_at_commit": head}, open(dst, "w"), indent=1)
print(dst, head)
