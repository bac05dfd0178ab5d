#!/bin/bash
# round 5: LayerNorm backward fused into the march weight-gradient kernel: MVIT_POOL_LNB_FUSE=0 never / 1 everywhere / 2 (default) stride 2 + small grids
for rep in 1 2; do
for shape in "8 4 8 28 28 1" "8 4 8 28 28 2" "8 2 8 56 56 1" "8 8 8 14 14 1"; do
  echo "separate: $(MVIT_POOL_LNB_FUSE=0 python3 tools/opbench.py poolbwd $shape 30 2>/dev/null | tail -1)"
  echo "fused   : $(MVIT_POOL_LNB_FUSE=1 python3 tools/opbench.py poolbwd $shape 30 2>/dev/null | tail -1)"
done; done
for rep in 1 2 3; do
  for v in 0 1 2; do
    echo "train step, MVIT_POOL_LNB_FUSE=$v: $(MVIT_POOL_LNB_FUSE=$v python bench.py --no-cpu-baseline --no-kernel-timing --no-forward-record --steps 30 --warmup 5 2>/dev/null | python -c 'import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(d["value"], d["ms_per_step"])')"
  done
done
