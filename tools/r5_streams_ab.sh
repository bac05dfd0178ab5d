#!/bin/bash
# round 5: HIP.STREAMS 2 vs 3 (vs uneven) for the inference forward, interleaved in one call (fp16 default arithmetic, B = 8 @448)
for rep in 1 2 3; do for st in 2 3; do
  echo "HIP.STREAMS $st fwd fp16: $(python bench.py --mode fwd --streams $st --no-cpu-baseline --no-kernel-timing --steps 60 --warmup 10 2>/dev/null | python -c 'import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(d["value"], d["ms_per_step"])')"
done; done
for st in 2 3; do echo "HIP.STREAMS $st window: $(python bench.py --mode window --streams $st --no-cpu-baseline --steps 3 --warmup 1 2>/dev/null | python -c 'import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(d["value"])')"; done
