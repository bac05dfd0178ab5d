#!/bin/bash
# kernel-only durations (rocprofv3) of opbench GEMM runs next to the event-timed numbers, ping-pong on / off
root=${GRAFT_REPO_ROOT:-$(pwd)}
cd /tmp && export TMPDIR=/tmp
for shp in "50176 1152 384 b" "50176 1536 384 bg" "50176 384 1536 br" "50176 384 384 br"; do
for pp in 1 0; do
  export MVIT_GEMM_PP=$pp
  rm -rf $root/gpurun_out/opb_prof
  out=$(rocprofv3 --kernel-trace --stats --output-format csv -d $root/gpurun_out/opb_prof -- python3 $root/tools/opbench.py gemm $shp 30 2>/dev/null | tail -1)
  echo "pp=$pp events: $out"
  python3 - <<PY
import csv, glob
f = glob.glob("$root/gpurun_out/opb_prof/**/*kernel_stats.csv", recursive=True)[0]
for r in csv.DictReader(open(f)):
    if "linear" in r["Name"]:
        print("      kernel-only: %-50s calls %4d avg %9.1f us" % (r["Name"][:50], int(r["Calls"]), float(r["AverageNs"]) / 1e3))
PY
done; done
rm -rf $root/gpurun_out/opb_prof
