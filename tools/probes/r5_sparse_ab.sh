#!/bin/bash
# round 5: strides >= 3 (k / v pooling of blocks 0-2): generic kernels (MVIT_POOL_SPARSE=0) vs the gather form, forward and backward op bench
for rep in 1 2; do
for shape in "8 1 8 112 112 8" "8 2 8 56 56 4" "8 2 8 56 56 8"; do
  echo "generic fwd: $(MVIT_POOL_SPARSE=0 python3 tools/opbench.py pool $shape 30 2>/dev/null | tail -1)"
  echo "gather  fwd: $(python3 tools/opbench.py pool $shape 30 2>/dev/null | tail -1)"
  echo "generic bwd: $(MVIT_POOL_SPARSE=0 python3 tools/opbench.py poolbwd $shape 30 2>/dev/null | tail -1)"
  echo "gather  bwd: $(python3 tools/opbench.py poolbwd $shape 30 2>/dev/null | tail -1)"
done; done
