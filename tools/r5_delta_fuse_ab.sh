#!/bin/bash
# round 5: delta + scaled queries produced by the dQ pass (MVIT_ATT_DELTA_FUSE=1, default) vs the separate delta kernel (0): op level and in the train step
for rep in 1 2; do
for shape in "8 4 6272 1568" "8 2 25088 1568" "8 8 1568 1568" "8 1 100352 1568"; do
  echo "separate: $(MVIT_ATT_DELTA_FUSE=0 python3 tools/opbench.py attnbwd $shape 20 2>/dev/null | tail -1)"
  echo "fused   : $(python3 tools/opbench.py attnbwd $shape 20 2>/dev/null | tail -1)"
done; done
for rep in 1 2 3; do
  for v in 0 1; do
    echo "train step, MVIT_ATT_DELTA_FUSE=$v: $(MVIT_ATT_DELTA_FUSE=$v python bench.py --no-cpu-baseline --no-kernel-timing --no-forward-record --steps 30 --warmup 5 2>/dev/null | python -c 'import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(d["value"], d["ms_per_step"])')"
  done
done
