#!/bin/bash
# round 5 (GPU box, repo root): long-key attention backward (Lk > 2048: blocks 1, 3, 14) as delta + (dQ || dK/dV on two streams) (default) vs
# sequential with the delta work inside the dQ pass (MVIT_ATT_BWD_SIDE=0), and side by side everywhere (=1): train step, interleaved
val() { python -c 'import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(d["value"], d["ms_per_step"])'; }
for rep in 1 2 3; do
  echo "train step, default            : $(python bench.py --no-cpu-baseline --no-kernel-timing --no-forward-record --steps 30 --warmup 5 2>/dev/null | val)"
  echo "train step, MVIT_ATT_BWD_SIDE=0: $(MVIT_ATT_BWD_SIDE=0 python bench.py --no-cpu-baseline --no-kernel-timing --no-forward-record --steps 30 --warmup 5 2>/dev/null | val)"
done
