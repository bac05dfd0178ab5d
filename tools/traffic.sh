#!/bin/bash
# usage (GPU box, repo root): tools/traffic.sh <kernel-substring> <outfile.json> [bench args]
# MARKER=<substring> (env): a kernel launched exactly once per CALL of the op (e.g. attn_bwd_delta for mvit_attention_bwd); the output
# then also carries traffic_bytes_per_call = all bytes of the matched kernels / number of marker launches.
# HBM-side bytes per launch of one kernel from the PMC counters, two separate passes (FETCH_SIZE costs 3 of the 4 TCC slots,
# WRITE_SIZE 2), corrected as MI355X_MICROARCH.md "HBM" prescribes: FETCH_SIZE is in KiB and on gfx950 tallies the 128-B
# requests of wide coalesced reads at 64 B -> doubled; WRITE_SIZE (KiB) is exact for 16-B-per-lane stores.
kern=$1; out=$2; shift 2
root=${GRAFT_REPO_ROOT:-$(pwd)}
cd /tmp && export TMPDIR=/tmp
rocprofv3 --pmc FETCH_SIZE --output-format csv -d $root/gpurun_out/traffic/f -- python3 $root/bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-kernel-timing --no-forward-record "$@" > /dev/null 2>&1
rocprofv3 --pmc WRITE_SIZE --output-format csv -d $root/gpurun_out/traffic/w -- python3 $root/bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-kernel-timing --no-forward-record "$@" > /dev/null 2>&1
python3 - <<PY
import csv, glob, json, collections
import os
marker = os.environ.get("MARKER", "")
calls = collections.Counter()
acc = collections.defaultdict(list)
for f in glob.glob("$root/gpurun_out/traffic/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        if "$kern" in r["Kernel_Name"]:
            acc[r["Counter_Name"]].append(float(r["Counter_Value"]))
            if marker and marker in r["Kernel_Name"]:
                calls[r["Counter_Name"]] += 1
fetch = sum(acc["FETCH_SIZE"]) / max(1, len(acc["FETCH_SIZE"])) * 1024.0 * 2.0
write = sum(acc["WRITE_SIZE"]) / max(1, len(acc["WRITE_SIZE"])) * 1024.0
d = {"kernel": "$kern", "args": "$*", "launches_sampled": len(acc["FETCH_SIZE"]), "fetch_bytes_per_launch": fetch,
     "write_bytes_per_launch": write, "traffic_bytes_per_launch": fetch + write,
     "method": "rocprofv3 --pmc FETCH_SIZE / --pmc WRITE_SIZE in separate passes; FETCH_SIZE KiB x2 (gfx950 128-B requests tallied at 64 B), WRITE_SIZE KiB x1"}
if marker and calls["FETCH_SIZE"] and calls["WRITE_SIZE"]:
    d["marker"] = marker
    d["calls_sampled"] = calls["FETCH_SIZE"]
    d["kernels_per_call"] = len(acc["FETCH_SIZE"]) / calls["FETCH_SIZE"]
    d["traffic_bytes_per_call"] = sum(acc["FETCH_SIZE"]) * 2048.0 / calls["FETCH_SIZE"] + sum(acc["WRITE_SIZE"]) * 1024.0 / calls["WRITE_SIZE"]
open("$root/$out", "w").write(json.dumps(d, indent=1) + "\n")
print(json.dumps(d))
PY
