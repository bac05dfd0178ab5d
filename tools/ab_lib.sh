#!/bin/bash
# A/B of two builds of the kernel library inside ONE gpurun call (boxes differ by a few percent, so never compare across calls):
#   tools/ab_lib.sh <tag> <opbench args...>   -> per-kernel average durations for lib/base_libmvit_hip.so and lib/libmvit_hip.so
tag=$1; shift
root=${GRAFT_REPO_ROOT:-$(pwd)}
cd /tmp && export TMPDIR=/tmp MVIT_NO_SIDE_STREAM=1
for which in base new base new; do
  if [ $which = base ]; then export MVIT_HIP_LIB=$root/aicity_action_amd/lib/base_libmvit_hip.so; else export MVIT_HIP_LIB=$root/aicity_action_amd/lib/libmvit_hip.so; fi
  rm -rf $root/gpurun_out/ab_$tag
  rocprofv3 --kernel-trace --stats --output-format csv -d $root/gpurun_out/ab_$tag -- python3 $root/tools/opbench.py "$@" > /dev/null 2>&1
  echo "== $which: $*"
  python3 - <<PY
import csv, glob
f = glob.glob("$root/gpurun_out/ab_$tag/**/*kernel_stats.csv", recursive=True)[0]
for r in csv.DictReader(open(f)):
    if float(r["TotalDurationNs"]) > 2e5 and "elementwise" not in r["Name"] and "distribution" not in r["Name"]:
        print("   %-60s calls %4d avg %9.1f us" % (r["Name"][:60], int(r["Calls"]), float(r["AverageNs"]) / 1e3))
PY
done
rm -rf $root/gpurun_out/ab_$tag
