#!/bin/bash
# round 5 probe (GPU box, repo root): stream priority of the library's side streams (weight gradients, side-by-side attention passes): default vs lowest vs highest
val() { python -c 'import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(d["value"], d["ms_per_step"])'; }
for rep in 1 2 3; do
  for p in 0 1 2; do
    echo "train step, MVIT_SIDE_PRIO=$p: $(MVIT_SIDE_PRIO=$p python bench.py --no-cpu-baseline --no-kernel-timing --no-forward-record --steps 30 --warmup 5 2>/dev/null | val)"
  done
done
