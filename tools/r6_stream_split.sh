#!/bin/bash
# GPU box: forward (fp16, B = 8 @448) with explicit sub-batch splits over the inference streams (MVIT_STREAM_SPLIT), interleaved
out=${1:-gpurun_out/r6_stream_split_ab.txt}
: > $out
run() { echo "split=$1 streams=$2 fwd fp16: $(MVIT_STREAM_SPLIT=$1 python bench.py --mode fwd --streams $2 --no-cpu-baseline --no-kernel-timing --steps 40 2>/dev/null | python -c 'import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(d["value"], d["ms_per_step"])')" >> $out; }
for rep in 1 2; do
  run "3,3,2" 3; run "2,3,3" 3; run "4,2,2" 3; run "2,2,4" 3; run "3,2,3" 3; run "4,4" 2; run "5,3" 2; run "3,5" 2; run "4,3,1" 3; run "2,2,2,2" 4; run "3,2,2,1" 4
done
cat $out
