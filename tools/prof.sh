#!/bin/bash
# usage (on the GPU box, from the repo root): tools/prof.sh <name> [bench args]   -> gpurun_out/<name>/
name=$1; shift
root=${GRAFT_REPO_ROOT:-$(pwd)}
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $root/gpurun_out/$name -- python3 $root/bench.py --steps 5 --warmup 2 --no-cpu-baseline --no-kernel-timing --no-forward-record "$@" > /dev/null 2>&1
python3 $root/tools/kstats.py $root/gpurun_out/$name 7
