#!/bin/bash
# round 5: pooling-conv backward op bench, 8-wide weight-gradient tiles vs the march form (same library, env switch), interleaved
for rep in 1 2; do
for shape in "8 4 8 28 28 1" "8 4 8 28 28 2" "8 2 8 56 56 1" "8 8 8 14 14 1" "8 1 8 112 112 1" "8 2 8 56 56 2"; do
  echo "tiled wgrad: $(MVIT_POOL_WGRAD_MARCH=0 python3 tools/opbench.py poolbwd $shape 30 2>/dev/null | tail -1)"
  echo "march wgrad: $(python3 tools/opbench.py poolbwd $shape 30 2>/dev/null | tail -1)"
done; done
