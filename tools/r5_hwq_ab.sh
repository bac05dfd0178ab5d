#!/bin/bash
# round 5 probe (GPU box, repo root): ROCm's GPU_MAX_HW_QUEUES (default 4) against the inference sub-batch streams
val() { python -c 'import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(d["value"], d["ms_per_step"])'; }
for rep in 1 2; do
for q in "" 2 8; do
  for st in 3 4; do
    echo "GPU_MAX_HW_QUEUES=${q:-default} HIP.STREAMS $st fwd fp16: $(env ${q:+GPU_MAX_HW_QUEUES=$q} python bench.py --mode fwd --streams $st --no-cpu-baseline --no-kernel-timing --steps 40 --warmup 8 2>/dev/null | val)"
  done
done
done
