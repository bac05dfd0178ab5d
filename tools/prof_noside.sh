#!/bin/bash
# kernel profile with the side streams off (per-kernel durations free of concurrency): tools/prof_noside.sh <name> [bench args]
name=$1; shift
root=${GRAFT_REPO_ROOT:-$(pwd)}
cd /tmp && export TMPDIR=/tmp MVIT_NO_SIDE_STREAM=1 MVIT_WGRAD_STREAM=0
rocprofv3 --kernel-trace --stats --output-format csv -d $root/gpurun_out/$name -- python3 $root/bench.py --steps 5 --warmup 2 --no-cpu-baseline --no-kernel-timing --no-forward-record --streams 1 "$@" > /dev/null 2>&1
python3 $root/tools/kstats.py $root/gpurun_out/$name 7
